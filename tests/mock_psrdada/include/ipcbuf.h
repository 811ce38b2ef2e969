/* TEST STAND-IN, not psrdada (see multilog.h in this directory).  The ipcbuf calls of the shim, from the reference's
 * call sites: /root/reference/src/process_baseband.cu:172 (ipcbuf_get_next_write), :199 (ipcbuf_mark_filled),
 * :807 (ipcbuf_get_next_read), :832 (ipcbuf_mark_cleared), :209,:309-310 (ipcbuf_get_nfull / _nbufs, on the data
 * block cast to ipcbuf_t*: the data block BEGINS with an ipcbuf_t). */
#ifndef MOCK_PSRDADA_IPCBUF_H
#define MOCK_PSRDADA_IPCBUF_H
#include <stdint.h>
#include <sys/types.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct ipcbuf_t {
    void *ctl;           /* the ring's control block (shared mapping) */
    char *base;          /* first buffer of this block */
    uint64_t bufsz, nbufs;
    int is_data;
} ipcbuf_t;
char *ipcbuf_get_next_read(ipcbuf_t *id, uint64_t *bytes);
int ipcbuf_mark_cleared(ipcbuf_t *id);
char *ipcbuf_get_next_write(ipcbuf_t *id);
int ipcbuf_mark_filled(ipcbuf_t *id, uint64_t nbytes);
uint64_t ipcbuf_get_nbufs(ipcbuf_t *id);
uint64_t ipcbuf_get_nfull(ipcbuf_t *id);
uint64_t ipcbuf_get_bufsz(ipcbuf_t *id);
#ifdef __cplusplus
}
#endif
#endif
