/* TEST INFRASTRUCTURE -- a stand-in for libpsrdada, NOT psrdada and not part of the product.
 *
 * It implements the psrdada symbols that vlite-fast_amd/csrc/pb_dada_shim.c calls (declared in ./include/, written
 * from the reference's call sites) so that the shim itself -- the real product file -- is compiled, linked and
 * executed: the mock sits UNDER the shim.  It pins nothing about psrdada's ABI or timing; a host with psrdada runs
 * tools/dada_selftest.sh for that.
 *
 * A ring (header block of 8 x 4096 bytes + data block of nbufs x bufsz bytes + a control block) is a file mapped
 * MAP_SHARED under $MOCK_PSRDADA_DIR (default /tmp), so that two PROCESSES can use it like a SysV psrdada ring: the
 * native process_baseband reading ring 0x40 while a Python writer fills it.  One writer, one reader per ring.
 * Blocking calls poll with a 50-us sleep and give up after $MOCK_PSRDADA_TIMEOUT_S (default 30) seconds, so that a
 * test that would hang fails instead.
 *
 * Data-block semantics (what the shim was written to, see include/ipcio.h): the writer fills buffers in order; a
 * buffer is handed over when full; end of data (dada_hdu_unlock_write) hands over the current buffer with whatever it
 * holds -- possibly 0 bytes -- flagged EOD.  The reader gets buffers in order; after the EOD buffer has been consumed
 * reads return 0 / open_block_read returns NULL until dada_hdu_unlock_read + dada_hdu_lock_read start the next
 * observation.
 *
 * Build (tests/test_dada_ring.py does this): gcc -shared -fPIC -Wall -Werror -Iinclude -o lib/libpsrdada.so
 * mock_psrdada.c -lpthread */
#define _GNU_SOURCE
#include <errno.h>
#include <fcntl.h>
#include <stdarg.h>
#include <stdatomic.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include "dada_hdu.h"

#define MOCK_MAGIC 0x6d6f636b64616461ull
#define MOCK_NHDR 8
#define MOCK_HDRSZ 4096
#define MOCK_MAXBUFS 256

typedef struct {
    uint64_t magic, bufsz, nbufs;
    _Atomic uint64_t hdr_filled, hdr_cleared;     /* header buffers posted / consumed */
    _Atomic uint64_t filled, cleared;             /* data buffers handed over / handed back */
    _Atomic int shutdown;
    uint64_t hdr_size[MOCK_NHDR];
    uint64_t size[MOCK_MAXBUFS];                  /* bytes in a handed-over buffer */
    int32_t eod[MOCK_MAXBUFS];
} mock_ctl;

static size_t ctl_bytes(void) { return (sizeof(mock_ctl) + 4095) & ~(size_t)4095; }

struct multilog_t {
    char name[64];
    FILE *fp[8];
    int nfp;
};

multilog_t *multilog_open(const char *program_name, char syslog_too)
{
    (void)syslog_too;
    multilog_t *m = (multilog_t *)calloc(1, sizeof *m);
    if (m) snprintf(m->name, sizeof m->name, "%s", program_name ? program_name : "");
    return m;
}

int multilog_add(multilog_t *m, FILE *fptr)
{
    if (!m || m->nfp >= 8) return -1;
    m->fp[m->nfp++] = fptr;
    return 0;
}

int multilog(multilog_t *m, int priority, const char *format, ...)
{
    (void)priority;
    if (!m) return -1;
    for (int i = 0; i < m->nfp; ++i) {
        va_list ap;
        va_start(ap, format);
        vfprintf(m->fp[i], format, ap);
        va_end(ap);
    }
    return 0;
}

int multilog_close(multilog_t *m)
{
    free(m);
    return 0;
}

static void ring_path(uint32_t key, char *out, size_t n)
{
    const char *d = getenv("MOCK_PSRDADA_DIR");
    snprintf(out, n, "%s/mock_psrdada_%08x.ring", d && *d ? d : "/tmp", key);
}

static double timeout_s(void)
{
    const char *t = getenv("MOCK_PSRDADA_TIMEOUT_S");
    return t && *t ? atof(t) : 30.0;
}

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

static void nap(void)
{
    struct timespec ts = {0, 50000};
    nanosleep(&ts, NULL);
}

/* ---- test control ---- */
int mock_psrdada_create(uint32_t key, uint64_t bufsz, uint64_t nbufs)
{
    if (!bufsz || !nbufs || nbufs > MOCK_MAXBUFS) return -1;
    char path[512];
    ring_path(key, path, sizeof path);
    int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0600);
    if (fd < 0) return -1;
    const size_t len = ctl_bytes() + (size_t)MOCK_NHDR * MOCK_HDRSZ + (size_t)(bufsz * nbufs);
    if (ftruncate(fd, (off_t)len) != 0) {
        close(fd);
        return -1;
    }
    mock_ctl *c = (mock_ctl *)mmap(NULL, ctl_bytes(), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (c == MAP_FAILED) return -1;
    memset(c, 0, sizeof *c);
    c->bufsz = bufsz;
    c->nbufs = nbufs;
    c->magic = MOCK_MAGIC;
    munmap(c, ctl_bytes());
    return 0;
}

int mock_psrdada_destroy(uint32_t key)
{
    char path[512];
    ring_path(key, path, sizeof path);
    return unlink(path);
}

static mock_ctl *map_ctl_only(uint32_t key)
{
    char path[512];
    ring_path(key, path, sizeof path);
    int fd = open(path, O_RDWR);
    if (fd < 0) return NULL;
    mock_ctl *c = (mock_ctl *)mmap(NULL, ctl_bytes(), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (c == MAP_FAILED) return NULL;
    if (c->magic != MOCK_MAGIC) {
        munmap(c, ctl_bytes());
        return NULL;
    }
    return c;
}

int mock_psrdada_shutdown(uint32_t key)
{
    mock_ctl *c = map_ctl_only(key);
    if (!c) return -1;
    atomic_store(&c->shutdown, 1);
    munmap(c, ctl_bytes());
    return 0;
}

int mock_psrdada_counts(uint32_t key, uint64_t *filled, uint64_t *cleared, uint64_t *hdr_filled, uint64_t *hdr_cleared)
{
    mock_ctl *c = map_ctl_only(key);
    if (!c) return -1;
    if (filled) *filled = atomic_load(&c->filled);
    if (cleared) *cleared = atomic_load(&c->cleared);
    if (hdr_filled) *hdr_filled = atomic_load(&c->hdr_filled);
    if (hdr_cleared) *hdr_cleared = atomic_load(&c->hdr_cleared);
    munmap(c, ctl_bytes());
    return 0;
}

/* ---- dada_hdu ---- */
dada_hdu_t *dada_hdu_create(multilog_t *log)
{
    dada_hdu_t *h = (dada_hdu_t *)calloc(1, sizeof *h);
    if (h) h->log = log;
    return h;
}

void dada_hdu_set_key(dada_hdu_t *hdu, key_t key)
{
    if (hdu) hdu->data_block_key = key;
}

int dada_hdu_connect(dada_hdu_t *hdu)
{
    if (!hdu || hdu->map) return -1;
    char path[512];
    ring_path((uint32_t)hdu->data_block_key, path, sizeof path);
    int fd = open(path, O_RDWR);
    if (fd < 0) return -1;
    struct stat st;
    if (fstat(fd, &st) != 0 || (size_t)st.st_size < ctl_bytes()) {
        close(fd);
        return -1;
    }
    void *m = mmap(NULL, (size_t)st.st_size, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return -1;
    mock_ctl *c = (mock_ctl *)m;
    if (c->magic != MOCK_MAGIC ||
        (size_t)st.st_size != ctl_bytes() + (size_t)MOCK_NHDR * MOCK_HDRSZ + (size_t)(c->bufsz * c->nbufs)) {
        munmap(m, (size_t)st.st_size);
        return -1;
    }
    hdu->map = m;
    hdu->map_len = (uint64_t)st.st_size;
    hdu->header_block = (ipcbuf_t *)calloc(1, sizeof(ipcbuf_t));
    hdu->data_block = (ipcio_t *)calloc(1, sizeof(ipcio_t));
    if (!hdu->header_block || !hdu->data_block) return -1;
    hdu->header_block->ctl = c;
    hdu->header_block->base = (char *)m + ctl_bytes();
    hdu->header_block->bufsz = MOCK_HDRSZ;
    hdu->header_block->nbufs = MOCK_NHDR;
    hdu->header_block->is_data = 0;
    hdu->data_block->buf.ctl = c;
    hdu->data_block->buf.base = (char *)m + ctl_bytes() + (size_t)MOCK_NHDR * MOCK_HDRSZ;
    hdu->data_block->buf.bufsz = c->bufsz;
    hdu->data_block->buf.nbufs = c->nbufs;
    hdu->data_block->buf.is_data = 1;
    return 0;
}

int dada_hdu_disconnect(dada_hdu_t *hdu)
{
    if (!hdu || !hdu->map) return -1;
    free(hdu->header_block);
    free(hdu->data_block);
    hdu->header_block = NULL;
    hdu->data_block = NULL;
    munmap(hdu->map, (size_t)hdu->map_len);
    hdu->map = NULL;
    return 0;
}

void dada_hdu_destroy(dada_hdu_t *hdu)
{
    if (!hdu) return;
    if (hdu->map) dada_hdu_disconnect(hdu);
    free(hdu);
}

int dada_hdu_lock_read(dada_hdu_t *hdu)
{
    if (!hdu || !hdu->map || hdu->data_block->rdwrt) return -1;
    hdu->data_block->rdwrt = 'r';
    hdu->data_block->eod_seen = 0;
    hdu->data_block->curbuf = NULL;
    hdu->data_block->bytes = 0;
    return 0;
}

static int data_hand_back(ipcio_t *ipc)
{
    mock_ctl *c = (mock_ctl *)ipc->buf.ctl;
    atomic_fetch_add(&c->cleared, 1);
    ipc->curbuf = NULL;
    ipc->bytes = 0;
    if (ipc->cur_is_eod) ipc->eod_seen = 1;
    return 0;
}

int dada_hdu_unlock_read(dada_hdu_t *hdu)
{
    if (!hdu || !hdu->map || hdu->data_block->rdwrt != 'r') return -1;
    if (hdu->data_block->curbuf) data_hand_back(hdu->data_block);       /* a buffer half read goes back */
    hdu->data_block->rdwrt = 0;
    return 0;
}

int dada_hdu_lock_write(dada_hdu_t *hdu)
{
    if (!hdu || !hdu->map || hdu->data_block->rdwrt) return -1;
    hdu->data_block->rdwrt = 'w';
    hdu->data_block->curbuf = NULL;
    hdu->data_block->bytes = 0;
    return 0;
}

/* the writer's next free data buffer (waits while the ring is full) */
static char *data_next_write(ipcio_t *ipc)
{
    mock_ctl *c = (mock_ctl *)ipc->buf.ctl;
    const double t0 = now_s();
    while (atomic_load(&c->filled) - atomic_load(&c->cleared) >= c->nbufs) {
        if (now_s() - t0 > timeout_s()) {
            fprintf(stderr, "mock_psrdada: ring full for %.0f s, giving up\n", timeout_s());
            return NULL;
        }
        nap();
    }
    return ipc->buf.base + (size_t)((atomic_load(&c->filled) % c->nbufs) * c->bufsz);
}

static void data_hand_over(ipcio_t *ipc, int eod)
{
    mock_ctl *c = (mock_ctl *)ipc->buf.ctl;
    const uint64_t slot = atomic_load(&c->filled) % c->nbufs;
    c->size[slot] = ipc->bytes;
    c->eod[slot] = eod;
    atomic_fetch_add(&c->filled, 1);          /* (seq_cst: size / eod are visible to whoever sees the count) */
    ipc->curbuf = NULL;
    ipc->bytes = 0;
}

int dada_hdu_unlock_write(dada_hdu_t *hdu)
{
    if (!hdu || !hdu->map || hdu->data_block->rdwrt != 'w') return -1;
    ipcio_t *ipc = hdu->data_block;
    if (!ipc->curbuf) {
        ipc->curbuf = data_next_write(ipc);
        ipc->bytes = 0;
        if (!ipc->curbuf) return -1;
    }
    data_hand_over(ipc, 1);                   /* end of data: the current buffer with whatever it holds */
    ipc->rdwrt = 0;
    return 0;
}

/* ---- ipcbuf (header block; nfull / nbufs also on the data block) ---- */
char *ipcbuf_get_next_read(ipcbuf_t *id, uint64_t *bytes)
{
    if (!id || id->is_data) return NULL;
    mock_ctl *c = (mock_ctl *)id->ctl;
    const double t0 = now_s();
    while (atomic_load(&c->hdr_filled) == atomic_load(&c->hdr_cleared)) {
        if (atomic_load(&c->shutdown)) return NULL;
        if (now_s() - t0 > timeout_s()) {
            fprintf(stderr, "mock_psrdada: no header for %.0f s, giving up\n", timeout_s());
            return NULL;
        }
        nap();
    }
    const uint64_t slot = atomic_load(&c->hdr_cleared) % MOCK_NHDR;
    if (bytes) *bytes = c->hdr_size[slot];
    return id->base + slot * MOCK_HDRSZ;
}

int ipcbuf_mark_cleared(ipcbuf_t *id)
{
    if (!id || id->is_data) return -1;
    mock_ctl *c = (mock_ctl *)id->ctl;
    if (atomic_load(&c->hdr_filled) == atomic_load(&c->hdr_cleared)) return -1;
    atomic_fetch_add(&c->hdr_cleared, 1);
    return 0;
}

char *ipcbuf_get_next_write(ipcbuf_t *id)
{
    if (!id || id->is_data) return NULL;
    mock_ctl *c = (mock_ctl *)id->ctl;
    const double t0 = now_s();
    while (atomic_load(&c->hdr_filled) - atomic_load(&c->hdr_cleared) >= MOCK_NHDR) {
        if (now_s() - t0 > timeout_s()) return NULL;
        nap();
    }
    return id->base + (atomic_load(&c->hdr_filled) % MOCK_NHDR) * MOCK_HDRSZ;
}

int ipcbuf_mark_filled(ipcbuf_t *id, uint64_t nbytes)
{
    if (!id || id->is_data || nbytes > MOCK_HDRSZ) return -1;
    mock_ctl *c = (mock_ctl *)id->ctl;
    c->hdr_size[atomic_load(&c->hdr_filled) % MOCK_NHDR] = nbytes;
    atomic_fetch_add(&c->hdr_filled, 1);
    return 0;
}

uint64_t ipcbuf_get_nbufs(ipcbuf_t *id) { return id ? id->nbufs : 0; }
uint64_t ipcbuf_get_bufsz(ipcbuf_t *id) { return id ? id->bufsz : 0; }

uint64_t ipcbuf_get_nfull(ipcbuf_t *id)
{
    if (!id) return 0;
    mock_ctl *c = (mock_ctl *)id->ctl;
    return id->is_data ? atomic_load(&c->filled) - atomic_load(&c->cleared)
                       : atomic_load(&c->hdr_filled) - atomic_load(&c->hdr_cleared);
}

/* ---- ipcio (data block) ---- */
ssize_t ipcio_write(ipcio_t *ipc, char *ptr, size_t bytes)
{
    if (!ipc || ipc->rdwrt != 'w') return -1;
    size_t done = 0;
    while (done < bytes) {
        if (!ipc->curbuf) {
            ipc->curbuf = data_next_write(ipc);
            ipc->bytes = 0;
            if (!ipc->curbuf) return (ssize_t)done;          /* short write: the ring stayed full */
        }
        size_t n = (size_t)(ipc->buf.bufsz - ipc->bytes);
        if (n > bytes - done) n = bytes - done;
        memcpy(ipc->curbuf + ipc->bytes, ptr + done, n);
        ipc->bytes += n;
        done += n;
        if (ipc->bytes == ipc->buf.bufsz) data_hand_over(ipc, 0);
    }
    return (ssize_t)done;
}

/* the reader's next handed-over buffer (waits for the writer) */
static int data_next_read(ipcio_t *ipc)
{
    mock_ctl *c = (mock_ctl *)ipc->buf.ctl;
    const double t0 = now_s();
    while (atomic_load(&c->filled) == atomic_load(&c->cleared)) {
        if (now_s() - t0 > timeout_s()) {
            fprintf(stderr, "mock_psrdada: no data for %.0f s, giving up\n", timeout_s());
            return -1;
        }
        nap();
    }
    const uint64_t slot = atomic_load(&c->cleared) % c->nbufs;
    ipc->curbuf = ipc->buf.base + (size_t)(slot * c->bufsz);
    ipc->curbufsz = c->size[slot];
    ipc->cur_is_eod = c->eod[slot];
    ipc->bytes = 0;
    return 0;
}

ssize_t ipcio_read(ipcio_t *ipc, char *ptr, size_t bytes)
{
    if (!ipc || ipc->rdwrt != 'r') return -1;
    size_t got = 0;
    while (got < bytes && !ipc->eod_seen) {
        if (!ipc->curbuf && data_next_read(ipc) != 0) return -1;
        size_t n = (size_t)(ipc->curbufsz - ipc->bytes);
        if (n > bytes - got) n = bytes - got;
        memcpy(ptr + got, ipc->curbuf + ipc->bytes, n);
        ipc->bytes += n;
        got += n;
        if (ipc->bytes == ipc->curbufsz) data_hand_back(ipc);
    }
    return (ssize_t)got;
}

char *ipcio_open_block_read(ipcio_t *ipc, uint64_t *bytes, uint64_t *block_id)
{
    if (!ipc || ipc->rdwrt != 'r') return NULL;
    if (ipc->curbuf || ipc->bytes) {
        fprintf(stderr, "mock_psrdada: ipcio_open_block_read with a buffer already open (ipcio_read in progress?)\n");
        return NULL;
    }
    if (ipc->eod_seen) return NULL;
    mock_ctl *c = (mock_ctl *)ipc->buf.ctl;
    const uint64_t id = atomic_load(&c->cleared);
    if (data_next_read(ipc) != 0) return NULL;
    if (bytes) *bytes = ipc->curbufsz;
    if (block_id) *block_id = id % c->nbufs;
    return ipc->curbuf;
}

ssize_t ipcio_close_block_read(ipcio_t *ipc, uint64_t bytes)
{
    if (!ipc || ipc->rdwrt != 'r' || !ipc->curbuf) return -1;
    if (bytes != ipc->curbufsz) {
        fprintf(stderr, "mock_psrdada: ipcio_close_block_read(%llu) of a %llu-byte buffer\n", (unsigned long long)bytes,
                (unsigned long long)ipc->curbufsz);
        return -1;
    }
    data_hand_back(ipc);
    return 0;
}
