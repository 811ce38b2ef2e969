/* TEST STAND-IN, not psrdada (see multilog.h in this directory).  The ipcio calls of the shim: ipcio_read
 * (/root/reference/src/process_baseband.cu:838,1034), ipcio_write (:324 via check_ipcio_write, :1418,1486,1491) and
 * psrdada's block-level pair ipcio_open_block_read / ipcio_close_block_read, which the reference does not use
 * (pb_dada_read_mt does: include/pb_dada.h) -- their contract here is the one the shim was written to:
 *   open:  the next filled buffer in place and its size; NULL once the observation's end-of-data buffer has been
 *          handed back; the end-of-data buffer itself may hold 0 bytes; never while ipcio_read has a buffer half read
 *   close: `bytes` must be the size open reported; the buffer goes back to the writer. */
#ifndef MOCK_PSRDADA_IPCIO_H
#define MOCK_PSRDADA_IPCIO_H
#include "ipcbuf.h"
#ifdef __cplusplus
extern "C" {
#endif
typedef struct ipcio_t {
    ipcbuf_t buf;        /* first member: (ipcbuf_t*) casts of the data block are the reference's habit */
    char rdwrt;          /* 'r', 'w' or 0 */
    char *curbuf;
    uint64_t curbufsz, bytes;
    int cur_is_eod, eod_seen;
} ipcio_t;
ssize_t ipcio_read(ipcio_t *ipc, char *ptr, size_t bytes);
ssize_t ipcio_write(ipcio_t *ipc, char *ptr, size_t bytes);
char *ipcio_open_block_read(ipcio_t *ipc, uint64_t *bytes, uint64_t *block_id);
ssize_t ipcio_close_block_read(ipcio_t *ipc, uint64_t bytes);
#ifdef __cplusplus
}
#endif
#endif
