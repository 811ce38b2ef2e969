/* psrdada shim: see include/pb_dada.h.  Needs psrdada's headers and libpsrdada; not built by default
 * (the build image has neither):  make -C vlite-fast_amd/csrc dada PSRDADA=<prefix>  */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dada_hdu.h"
#include "ipcbuf.h"
#include "ipcio.h"
#include "multilog.h"

#include "pb_dada.h"

struct pb_dada {
    dada_hdu_t *hdu;
    multilog_t *log;
    int mode;
    int locked;
};

pb_dada *pb_dada_open(uint32_t key, int mode, char *err, uint64_t errlen)
{
    pb_dada *d = (pb_dada *)calloc(1, sizeof *d);
    if (!d) return NULL;
    d->mode = mode;
    d->log = multilog_open("process_baseband", 0);
    multilog_add(d->log, stderr);
    d->hdu = dada_hdu_create(d->log);                       /* :541 */
    dada_hdu_set_key(d->hdu, (key_t)key);                   /* :542 */
    if (dada_hdu_connect(d->hdu) != 0) {                    /* :543 */
        if (err) snprintf(err, errlen, "Unable to connect to PSRDADA buffer key=%x!", key);
        dada_hdu_destroy(d->hdu);
        multilog_close(d->log);
        free(d);
        return NULL;
    }
    return d;
}

int64_t pb_dada_next_header(pb_dada *d, char *dst)
{
    if (!d || d->mode != PB_DADA_READ) return -1;
    if (!d->locked) {
        if (dada_hdu_lock_read(d->hdu) < 0) return -2;     /* :799 */
        d->locked = 1;
    }
    uint64_t hdr_size = 0;
    char *ascii_hdr = ipcbuf_get_next_read(d->hdu->header_block, &hdr_size);     /* :807 (blocks) */
    if (!ascii_hdr) return 0;                                                    /* :810: shut down, or an error */
    if (hdr_size > PB_DADA_HDR_SIZE) hdr_size = PB_DADA_HDR_SIZE;
    memcpy(dst, ascii_hdr, hdr_size);                                            /* :831 */
    if (ipcbuf_mark_cleared(d->hdu->header_block) < 0) return -3;                /* :832 */
    return (int64_t)hdr_size;
}

int64_t pb_dada_read(pb_dada *d, void *buf, uint64_t nbytes)
{
    if (!d || d->mode != PB_DADA_READ) return -1;
    return (int64_t)ipcio_read(d->hdu->data_block, (char *)buf, nbytes);         /* :838, :1034 */
}

int pb_dada_end_read(pb_dada *d)
{
    if (!d || d->mode != PB_DADA_READ) return -1;
    if (!d->locked) return 0;
    d->locked = 0;
    return dada_hdu_unlock_read(d->hdu);                                         /* :1513 */
}

int pb_dada_write_header(pb_dada *d, const char *hdr)
{
    if (!d || d->mode != PB_DADA_WRITE) return -1;
    if (!d->locked) {
        if (dada_hdu_lock_write(d->hdu) < 0) return -2;                          /* :172 */
        d->locked = 1;
    }
    char *dst = ipcbuf_get_next_write(d->hdu->header_block);                     /* :174 */
    if (!dst) return -3;
    memcpy(dst, hdr, PB_DADA_HDR_SIZE);
    return ipcbuf_mark_filled(d->hdu->header_block, PB_DADA_HDR_SIZE);           /* :199 */
}

int64_t pb_dada_write(pb_dada *d, const void *buf, uint64_t nbytes)
{
    if (!d || d->mode != PB_DADA_WRITE || !d->locked) return -1;
    return (int64_t)ipcio_write(d->hdu->data_block, (char *)buf, nbytes);        /* :1418, :1486, :1491 */
}

int pb_dada_end_write(pb_dada *d)
{
    if (!d || d->mode != PB_DADA_WRITE) return -1;
    if (!d->locked) return 0;
    d->locked = 0;
    return dada_hdu_unlock_write(d->hdu);                                        /* :1501, :1508 */
}

void pb_dada_close(pb_dada *d)
{
    if (!d) return;
    if (d->locked) {
        if (d->mode == PB_DADA_READ) dada_hdu_unlock_read(d->hdu);
        else dada_hdu_unlock_write(d->hdu);
    }
    dada_hdu_disconnect(d->hdu);
    dada_hdu_destroy(d->hdu);
    multilog_close(d->log);
    free(d);
}
