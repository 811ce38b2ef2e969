/* psrdada shim: see include/pb_dada.h.  Needs psrdada's headers and libpsrdada; not built by default
 * (the build image has neither):  make -C vlite-fast_amd/csrc dada PSRDADA=<prefix>  */
#include <pthread.h>
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
    /* block-level reading (pb_dada_read_mt): the filled buffer that is open, and how far it has been consumed */
    char *blk;
    uint64_t blk_size, blk_pos;
    int how;        /* 0: nothing read yet in this observation, 1: ipcio_read, 2: block level */
};

typedef struct {
    char *dst;
    const char *src;
    size_t n;
} pb_copy_job;

static void *pb_copy_thread(void *p)
{
    pb_copy_job *j = (pb_copy_job *)p;
    memcpy(j->dst, j->src, j->n);
    return NULL;
}

/* memcpy split over nthreads threads (page-aligned pieces); small copies stay on the caller's thread */
static void pb_copy_mt(char *dst, const char *src, size_t n, int nthreads)
{
    if (nthreads > 16) nthreads = 16;
    if (nthreads < 2 || n < ((size_t)8 << 20)) {
        memcpy(dst, src, n);
        return;
    }
    pthread_t th[16];
    pb_copy_job job[16];
    int started[16];
    size_t per = ((n / (size_t)nthreads) + 4095) & ~(size_t)4095, off = 0;
    int k = 0;
    for (; k < nthreads && off < n; ++k) {
        job[k].dst = dst + off;
        job[k].src = src + off;
        job[k].n = n - off < per ? n - off : per;
        off += job[k].n;
        started[k] = pthread_create(&th[k], NULL, pb_copy_thread, &job[k]) == 0;
        if (!started[k]) pb_copy_thread(&job[k]);
    }
    for (int i = 0; i < k; ++i)
        if (started[i]) pthread_join(th[i], NULL);
}

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
    if (d->how == 2) return -4;                                                  /* block-level reads are in progress */
    d->how = 1;
    return (int64_t)ipcio_read(d->hdu->data_block, (char *)buf, nbytes);         /* :838, :1034 */
}

int64_t pb_dada_read_mt(pb_dada *d, void *buf, uint64_t nbytes, int nthreads)
{
    if (!d || d->mode != PB_DADA_READ) return -1;
    if (d->how == 1) return -4;                                                  /* ipcio_read is in progress */
    d->how = 2;
    uint64_t got = 0;
    while (got < nbytes) {
        if (!d->blk) {
            uint64_t sz = 0, id = 0;
            char *p = ipcio_open_block_read(d->hdu->data_block, &sz, &id);       /* next filled buffer, in place */
            if (!p || sz == 0) {                                                 /* end of data */
                if (p) ipcio_close_block_read(d->hdu->data_block, 0);
                break;
            }
            d->blk = p;
            d->blk_size = sz;
            d->blk_pos = 0;
        }
        uint64_t n = d->blk_size - d->blk_pos;
        if (n > nbytes - got) n = nbytes - got;
        pb_copy_mt((char *)buf + got, d->blk + d->blk_pos, (size_t)n, nthreads);
        got += n;
        d->blk_pos += n;
        if (d->blk_pos == d->blk_size) {                                         /* all of it consumed: hand it back */
            if (ipcio_close_block_read(d->hdu->data_block, d->blk_size) < 0) return -3;
            d->blk = NULL;
        }
    }
    return (int64_t)got;
}

int pb_dada_end_read(pb_dada *d)
{
    if (!d || d->mode != PB_DADA_READ) return -1;
    if (!d->locked) return 0;
    if (d->blk) {                                                                /* a buffer still open: hand it back */
        ipcio_close_block_read(d->hdu->data_block, d->blk_size);
        d->blk = NULL;
    }
    d->how = 0;
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
        if (d->mode == PB_DADA_READ && d->blk) ipcio_close_block_read(d->hdu->data_block, d->blk_size);
        if (d->mode == PB_DADA_READ) dada_hdu_unlock_read(d->hdu);
        else dada_hdu_unlock_write(d->hdu);
    }
    dada_hdu_disconnect(d->hdu);
    dada_hdu_destroy(d->hdu);
    multilog_close(d->log);
    free(d);
}
