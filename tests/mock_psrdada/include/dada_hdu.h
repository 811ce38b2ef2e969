/* TEST STAND-IN, not psrdada (see multilog.h in this directory).  The dada_hdu calls of the shim, from
 * /root/reference/src/process_baseband.cu:541-543 (create / set_key / connect), :75-76 (disconnect / destroy),
 * :170,:315 (lock_write / unlock_write), :799,:222 (lock_read / unlock_read) and the two members the reference
 * reaches into (hdu->header_block, hdu->data_block: :172,:214). */
#ifndef MOCK_PSRDADA_DADA_HDU_H
#define MOCK_PSRDADA_DADA_HDU_H
#include "ipcio.h"
#include "multilog.h"
#ifdef __cplusplus
extern "C" {
#endif
typedef struct dada_hdu_t {
    multilog_t *log;
    key_t data_block_key;
    ipcbuf_t *header_block;
    ipcio_t *data_block;
    void *map;           /* the ring's mapping and its length */
    uint64_t map_len;
} dada_hdu_t;
dada_hdu_t *dada_hdu_create(multilog_t *log);
void dada_hdu_set_key(dada_hdu_t *hdu, key_t key);
int dada_hdu_connect(dada_hdu_t *hdu);
int dada_hdu_disconnect(dada_hdu_t *hdu);
void dada_hdu_destroy(dada_hdu_t *hdu);
int dada_hdu_lock_read(dada_hdu_t *hdu);
int dada_hdu_unlock_read(dada_hdu_t *hdu);
int dada_hdu_lock_write(dada_hdu_t *hdu);
int dada_hdu_unlock_write(dada_hdu_t *hdu);

/* ---- test control (what dada_db does on a real host): not psrdada API ---- */
int mock_psrdada_create(uint32_t key, uint64_t bufsz, uint64_t nbufs);    /* dada_db -k KEY -b bufsz -n nbufs */
int mock_psrdada_destroy(uint32_t key);                                   /* dada_db -k KEY -d */
int mock_psrdada_shutdown(uint32_t key);          /* a header reader with nothing pending gets NULL instead of waiting */
int mock_psrdada_counts(uint32_t key, uint64_t *filled, uint64_t *cleared, uint64_t *hdr_filled, uint64_t *hdr_cleared);
#ifdef __cplusplus
}
#endif
#endif
