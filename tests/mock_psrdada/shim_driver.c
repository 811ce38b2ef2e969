/* TEST INFRASTRUCTURE: drives vlite-fast_amd/csrc/pb_dada_shim.c (the product file, compiled into this program with
 * -fsanitize=address,undefined) over the psrdada stand-in of this directory.  tests/test_dada_ring.py builds and runs
 * it; exit status 0 and "shim_driver: ok" = every check passed with no sanitizer report.
 * Cases: connect failure; header + data through ipcio_read; an observation whose length is a multiple of the buffer
 * size (empty end-of-data buffer) and one that is not, through the block-level interface with odd request sizes; a
 * stream larger than the ring written by a second thread while this one reads it with the multi-threaded copy;
 * refusal to mix the two read interfaces; shutdown; close with a lock held and a buffer open. */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dada_hdu.h"
#include "pb_dada.h"

#define CHECK(c)                                                                  \
    do {                                                                          \
        if (!(c)) {                                                               \
            fprintf(stderr, "shim_driver: %s:%d: %s failed\n", __FILE__, __LINE__, #c); \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

static unsigned char byte_at(uint64_t i, unsigned seed) { return (unsigned char)((i * 2654435761u + seed) >> 7); }

static void fill(unsigned char *p, uint64_t off, uint64_t n, unsigned seed)
{
    for (uint64_t i = 0; i < n; ++i) p[i] = byte_at(off + i, seed);
}

static void check_stream(const unsigned char *p, uint64_t off, uint64_t n, unsigned seed)
{
    for (uint64_t i = 0; i < n; ++i)
        if (p[i] != byte_at(off + i, seed)) {
            fprintf(stderr, "shim_driver: byte %llu of the stream differs\n", (unsigned long long)(off + i));
            exit(1);
        }
}

static void make_header(char *h, const char *name)
{
    memset(h, 0, PB_DADA_HDR_SIZE);
    snprintf(h, PB_DADA_HDR_SIZE, "NAME %s\nSTATIONID 7\n", name);
}

typedef struct {
    uint32_t key;
    uint64_t total;
    unsigned seed;
} feed_job;

static void *feeder(void *arg)
{
    feed_job *j = (feed_job *)arg;
    char err[256];
    pb_dada *w = pb_dada_open(j->key, PB_DADA_WRITE, err, sizeof err);
    CHECK(w);
    char hdr[PB_DADA_HDR_SIZE];
    make_header(hdr, "big");
    CHECK(pb_dada_write_header(w, hdr) == 0);
    const uint64_t piece = 5 * 1000 * 1000 + 17;
    unsigned char *buf = (unsigned char *)malloc(piece);
    CHECK(buf);
    for (uint64_t off = 0; off < j->total; off += piece) {
        const uint64_t n = j->total - off < piece ? j->total - off : piece;
        fill(buf, off, n, j->seed);
        CHECK(pb_dada_write(w, buf, n) == (int64_t)n);
    }
    free(buf);
    CHECK(pb_dada_end_write(w) == 0);
    pb_dada_close(w);
    return NULL;
}

int main(void)
{
    char err[256] = "";
    char hdr[PB_DADA_HDR_SIZE], got_hdr[PB_DADA_HDR_SIZE];
    const uint32_t KEY = 0x5d01, BIG = 0x5d02;

    /* no such ring */
    CHECK(pb_dada_open(0x5dff, PB_DADA_READ, err, sizeof err) == NULL);
    CHECK(strstr(err, "key=5dff") != NULL);

    CHECK(mock_psrdada_create(KEY, 1000, 4) == 0);
    pb_dada *w = pb_dada_open(KEY, PB_DADA_WRITE, err, sizeof err);
    pb_dada *r = pb_dada_open(KEY, PB_DADA_READ, err, sizeof err);
    CHECK(w && r);
    unsigned char buf[4000], out[4000];

    /* wrong-mode and out-of-order calls are refused */
    CHECK(pb_dada_write(w, buf, 8) < 0);                       /* data before a header */
    CHECK(pb_dada_read(w, buf, 8) < 0 && pb_dada_write(r, buf, 8) < 0);
    CHECK(pb_dada_end_write(w) == 0 && pb_dada_end_read(r) == 0);   /* nothing locked: no-ops */

    /* observation 1: 2500 bytes (two full buffers + a partial end-of-data buffer), ipcio_read */
    make_header(hdr, "obs1");
    CHECK(pb_dada_write_header(w, hdr) == 0);
    fill(buf, 0, 2500, 1);
    CHECK(pb_dada_write(w, buf, 2500) == 2500);
    CHECK(pb_dada_end_write(w) == 0);
    CHECK(pb_dada_next_header(r, got_hdr) == PB_DADA_HDR_SIZE && memcmp(hdr, got_hdr, PB_DADA_HDR_SIZE) == 0);
    CHECK(pb_dada_read(r, out, 700) == 700);
    CHECK(pb_dada_read_mt(r, out, 10, 4) == -4);               /* the two interfaces do not mix */
    CHECK(pb_dada_read(r, out + 700, 3000) == 1800);           /* short at end of data */
    check_stream(out, 0, 2500, 1);
    CHECK(pb_dada_read(r, out, 100) == 0);
    CHECK(pb_dada_end_read(r) == 0);

    /* observation 2: exactly two buffers (the end-of-data buffer is empty), block level, odd request sizes */
    make_header(hdr, "obs2");
    CHECK(pb_dada_write_header(w, hdr) == 0);
    fill(buf, 0, 2000, 2);
    CHECK(pb_dada_write(w, buf, 2000) == 2000);
    CHECK(pb_dada_end_write(w) == 0);
    CHECK(pb_dada_next_header(r, got_hdr) == PB_DADA_HDR_SIZE && strstr(got_hdr, "obs2"));
    uint64_t got = 0;
    for (;;) {
        int64_t n = pb_dada_read_mt(r, out + got, 333, 3);
        CHECK(n >= 0);
        got += (uint64_t)n;
        if (n < 333) break;
    }
    CHECK(got == 2000);
    check_stream(out, 0, 2000, 2);
    CHECK(pb_dada_read_mt(r, out, 50, 1) == 0);
    CHECK(pb_dada_read(r, out, 10) == -4);
    CHECK(pb_dada_end_read(r) == 0);
    uint64_t f = 0, c = 0, hf = 0, hc = 0;
    CHECK(mock_psrdada_counts(KEY, &f, &c, &hf, &hc) == 0);
    CHECK(f == 6 && c == 6 && hf == 2 && hc == 2);            /* every buffer was handed back, both headers cleared */

    /* observation 3: 2300 bytes, block level, ended by the reader after 1100 (a buffer open and half consumed) */
    make_header(hdr, "obs3");
    CHECK(pb_dada_write_header(w, hdr) == 0);
    fill(buf, 0, 2300, 3);
    CHECK(pb_dada_write(w, buf, 2300) == 2300);
    CHECK(pb_dada_end_write(w) == 0);
    CHECK(pb_dada_next_header(r, got_hdr) == PB_DADA_HDR_SIZE);
    CHECK(pb_dada_read_mt(r, out, 1100, 2) == 1100);
    check_stream(out, 0, 1100, 3);
    pb_dada_close(r);                                          /* locked, a block open: both are released */
    CHECK(mock_psrdada_counts(KEY, &f, &c, NULL, NULL) == 0 && c == 8);
    pb_dada_close(w);

    /* shutdown: a header reader with nothing pending gets "ring closed" */
    r = pb_dada_open(KEY, PB_DADA_READ, err, sizeof err);
    CHECK(r && mock_psrdada_shutdown(KEY) == 0);
    CHECK(pb_dada_next_header(r, got_hdr) == 0);
    pb_dada_close(r);
    CHECK(mock_psrdada_destroy(KEY) == 0);

    /* a stream four times the ring, written by another thread; 20-MB reads copied by 8 threads (pieces >= 8 MiB) */
    CHECK(mock_psrdada_create(BIG, 12u << 20, 3) == 0);
    feed_job job = {BIG, (uint64_t)150 * 1000 * 1000 + 123, 9};
    pthread_t th;
    CHECK(pthread_create(&th, NULL, feeder, &job) == 0);
    r = pb_dada_open(BIG, PB_DADA_READ, err, sizeof err);
    CHECK(r);
    CHECK(pb_dada_next_header(r, got_hdr) == PB_DADA_HDR_SIZE && strstr(got_hdr, "big"));
    const uint64_t req = 20 * 1000 * 1000;
    unsigned char *big = (unsigned char *)malloc(req);
    CHECK(big);
    got = 0;
    for (;;) {
        int64_t n = pb_dada_read_mt(r, big, req, 8);
        CHECK(n >= 0);
        check_stream(big, got, (uint64_t)n, 9);
        got += (uint64_t)n;
        if ((uint64_t)n < req) break;
    }
    CHECK(got == job.total);
    CHECK(pb_dada_end_read(r) == 0);
    pthread_join(th, NULL);
    free(big);
    pb_dada_close(r);
    CHECK(mock_psrdada_destroy(BIG) == 0);
    printf("shim_driver: ok\n");
    return 0;
}
