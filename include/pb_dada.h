/*
 * pb_dada.h -- flat C face of the psrdada calls the reference's process_baseband makes on its three
 * rings, so that the Python host can bind them with ctypes (psrdada's own structures -- dada_hdu_t,
 * ipcio_t, ipcbuf_t, multilog_t -- never cross the boundary).
 *
 * Implementation: vlite-fast_amd/csrc/pb_dada_shim.c, compiled ONLY where psrdada is installed
 * (`make -C vlite-fast_amd/csrc dada PSRDADA=/path/to/psrdada/prefix`); it is a few lines around
 * exactly these calls of /root/reference/src/process_baseband.cu:
 *   connect        :541-569   dada_hdu_create / dada_hdu_set_key / dada_hdu_connect
 *   header read    :799-837   dada_hdu_lock_read, ipcbuf_get_next_read (header_block), ipcbuf_mark_cleared
 *   data read      :838, :1034 ipcio_read (data_block)
 *   data read, block level (pb_dada_read_mt): the same bytes through psrdada's own zero-copy interface to the
 *                  data block, ipcio_open_block_read / ipcio_close_block_read (ipcio.h), each filled buffer
 *                  copied out by several threads -- ipcio_read is one memcpy thread, ~9 GB/s = ~35x real
 *                  time for 257.6 MB per second of data, :1034-1038
 *   end of obs     :1513      dada_hdu_unlock_read
 *   header write   :172-199, :981-989  dada_hdu_lock_write, ipcbuf_get_next_write, ipcbuf_mark_filled (4096)
 *   data write     :1416-1422 (coadd ring, one segment), :1482-1494 (10 s, then 1 s)  ipcio_write
 *   end of obs     :1498-1511 dada_hdu_unlock_write
 * Where psrdada is absent the tests compile the real shim (vlite-fast_amd/csrc/pb_dada_shim.c) against
 * tests/mock_psrdada -- declarations of exactly the psrdada calls above and shared-file rings behind them (test
 * infrastructure: it sits UNDER the shim and pins nothing about psrdada's ABI).
 */
#ifndef PB_DADA_H
#define PB_DADA_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pb_dada pb_dada;

#define PB_DADA_READ 0
#define PB_DADA_WRITE 1
#define PB_DADA_HDR_SIZE 4096

/* Connect to the HDU with this SysV key (as -k/-K/-C give it, hexadecimal on the command line).
 * NULL on failure, with a message in err. */
pb_dada *pb_dada_open(uint32_t key, int mode, char *err, uint64_t errlen);
/* Reader: lock the HDU for reading and block until the writer has posted an observation header;
 * copies it (up to PB_DADA_HDR_SIZE bytes) and clears the header buffer.  Returns the header size,
 * 0 when the ring has been shut down (ipcbuf_get_next_read returned NULL), < 0 on error. */
int64_t pb_dada_next_header(pb_dada *d, char *dst);
/* Reader: ipcio_read.  Returns bytes read; 0 at end of data; < 0 on error. */
int64_t pb_dada_read(pb_dada *d, void *buf, uint64_t nbytes);
/* Reader, block level: same contract as pb_dada_read (fills buf with up to nbytes bytes of the data stream, returns
 * the count, 0 at end of data, < 0 on error), but the bytes come straight out of the ring's filled buffers
 * (ipcio_open_block_read; a buffer is handed back with ipcio_close_block_read once all of it has been consumed,
 * a partly consumed one stays open inside the handle between calls) and pieces of 8 MiB or more are copied by
 * nthreads threads (1 = plain memcpy).  A handle reads EITHER with pb_dada_read OR with pb_dada_read_mt during
 * one observation: psrdada's ipcio keeps its own position for ipcio_read and refuses to mix the two. */
int64_t pb_dada_read_mt(pb_dada *d, void *buf, uint64_t nbytes, int nthreads);
/* Reader: the observation is over (dada_hdu_unlock_read). */
int pb_dada_end_read(pb_dada *d);
/* Writer: lock the HDU for writing and post a PB_DADA_HDR_SIZE-byte header. */
int pb_dada_write_header(pb_dada *d, const char *hdr);
/* Writer: ipcio_write.  Returns bytes written or < 0. */
int64_t pb_dada_write(pb_dada *d, const void *buf, uint64_t nbytes);
/* Writer: end of data for this observation (dada_hdu_unlock_write); a no-op if no header was posted. */
int pb_dada_end_write(pb_dada *d);
/* dada_hdu_disconnect + dada_hdu_destroy */
void pb_dada_close(pb_dada *d);

#ifdef __cplusplus
}
#endif
#endif
