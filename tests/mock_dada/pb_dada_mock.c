/* TEST INFRASTRUCTURE -- not psrdada and not part of the product.
 * Exports the symbols of include/pb_dada.h over rings that live in this shared object's memory, so that
 * vlite-fast_amd/dada.PsrdadaRing (ctypes binding + call order) can be exercised where psrdada is absent.
 * It enforces the protocol the real shim relies on: a header before data, data only between header and
 * end-of-data, a reader sees observations in order and 0 bytes at the end of each.
 * Build: gcc -shared -fPIC -Iinclude -o tests/mock_dada/libpb_dada_mock.so tests/mock_dada/pb_dada_mock.c */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pb_dada.h"

#define MAXRING 8
#define MAXOBS 16

typedef struct {
    char hdr[PB_DADA_HDR_SIZE];
    unsigned char *data;
    uint64_t len, cap;
    int closed;
} mock_obs;

typedef struct {
    uint32_t key;
    int used;
    mock_obs obs[MAXOBS];
    int nobs;         /* observations posted */
    int robs;         /* observation the reader is on */
    uint64_t rpos;
    int shutdown;
} mock_ring;

static mock_ring g_rings[MAXRING];

struct pb_dada {
    mock_ring *r;
    int mode;
    int in_obs;
    int how;              /* 0 none, 1 pb_dada_read, 2 pb_dada_read_mt: psrdada refuses to mix them */
    uint64_t blocks_opened, blocks_closed;
};

static uint64_t g_block_size = 1 << 20;   /* the mock ring's buffer size (dada_db -b) for block-level reads */
void pb_dada_mock_set_block_size(uint64_t n) { g_block_size = n ? n : 1; }

/* test control: create / destroy a ring (dada_db -k KEY ... / dada_db -d), mark it shut down */
int pb_dada_mock_create(uint32_t key)
{
    for (int i = 0; i < MAXRING; ++i)
        if (!g_rings[i].used) {
            memset(&g_rings[i], 0, sizeof g_rings[i]);
            g_rings[i].used = 1;
            g_rings[i].key = key;
            return 0;
        }
    return -1;
}

static mock_ring *find(uint32_t key)
{
    for (int i = 0; i < MAXRING; ++i)
        if (g_rings[i].used && g_rings[i].key == key) return &g_rings[i];
    return NULL;
}

void pb_dada_mock_destroy(uint32_t key)
{
    mock_ring *r = find(key);
    if (!r) return;
    for (int i = 0; i < r->nobs; ++i) free(r->obs[i].data);
    r->used = 0;
}

void pb_dada_mock_shutdown(uint32_t key)
{
    mock_ring *r = find(key);
    if (r) r->shutdown = 1;
}

pb_dada *pb_dada_open(uint32_t key, int mode, char *err, uint64_t errlen)
{
    mock_ring *r = find(key);
    if (!r) {
        if (err) snprintf(err, errlen, "Unable to connect to PSRDADA buffer key=%x!", key);
        return NULL;
    }
    pb_dada *d = (pb_dada *)calloc(1, sizeof *d);
    d->r = r;
    d->mode = mode;
    return d;
}

int64_t pb_dada_next_header(pb_dada *d, char *dst)
{
    if (!d || d->mode != PB_DADA_READ) return -1;
    if (d->in_obs) return -4;                         /* the previous observation was not ended */
    mock_ring *r = d->r;
    if (r->robs >= r->nobs) return r->shutdown ? 0 : -5;   /* the real call would block here */
    memcpy(dst, r->obs[r->robs].hdr, PB_DADA_HDR_SIZE);
    r->rpos = 0;
    d->in_obs = 1;
    return PB_DADA_HDR_SIZE;
}

/* block-level read: the stream is handed out buffer by buffer (g_block_size bytes each, the last one short),
 * a buffer is "closed" exactly when its last byte has been consumed */
int64_t pb_dada_read_mt(pb_dada *d, void *buf, uint64_t nbytes, int nthreads)
{
    if (!d || d->mode != PB_DADA_READ || !d->in_obs) return -1;
    if (d->how == 1) return -4;
    if (nthreads < 1) return -2;
    d->how = 2;
    mock_obs *o = &d->r->obs[d->r->robs];
    uint64_t got = 0;
    while (got < nbytes && d->r->rpos < o->len) {
        const uint64_t bstart = d->r->rpos / g_block_size * g_block_size;
        uint64_t bend = bstart + g_block_size;
        if (bend > o->len) bend = o->len;
        if (d->r->rpos == bstart) d->blocks_opened++;
        uint64_t n = bend - d->r->rpos;
        if (n > nbytes - got) n = nbytes - got;
        memcpy((char *)buf + got, o->data + d->r->rpos, n);
        got += n;
        d->r->rpos += n;
        if (d->r->rpos == bend) d->blocks_closed++;
    }
    return (int64_t)got;
}

/* test probe: buffers opened / handed back by block-level reads on this handle */
void pb_dada_mock_block_counts(pb_dada *d, uint64_t *opened, uint64_t *closed)
{
    *opened = d->blocks_opened;
    *closed = d->blocks_closed;
}

int64_t pb_dada_read(pb_dada *d, void *buf, uint64_t nbytes)
{
    if (!d || d->mode != PB_DADA_READ || !d->in_obs) return -1;
    if (d->how == 2) return -4;
    d->how = 1;
    mock_obs *o = &d->r->obs[d->r->robs];
    uint64_t left = o->len - d->r->rpos;
    if (nbytes > left) nbytes = left;
    memcpy(buf, o->data + d->r->rpos, nbytes);
    d->r->rpos += nbytes;
    return (int64_t)nbytes;
}

int pb_dada_end_read(pb_dada *d)
{
    if (!d || d->mode != PB_DADA_READ) return -1;
    if (!d->in_obs) return 0;
    d->in_obs = 0;
    d->how = 0;
    d->r->robs++;
    return 0;
}

int pb_dada_write_header(pb_dada *d, const char *hdr)
{
    if (!d || d->mode != PB_DADA_WRITE) return -1;
    if (d->in_obs) return -4;
    mock_ring *r = d->r;
    if (r->nobs >= MAXOBS) return -6;
    mock_obs *o = &r->obs[r->nobs++];
    memset(o, 0, sizeof *o);
    memcpy(o->hdr, hdr, PB_DADA_HDR_SIZE);
    d->in_obs = 1;
    return 0;
}

int64_t pb_dada_write(pb_dada *d, const void *buf, uint64_t nbytes)
{
    if (!d || d->mode != PB_DADA_WRITE || !d->in_obs) return -1;
    mock_obs *o = &d->r->obs[d->r->nobs - 1];
    if (o->len + nbytes > o->cap) {
        o->cap = (o->len + nbytes) * 2;
        o->data = (unsigned char *)realloc(o->data, o->cap);
    }
    memcpy(o->data + o->len, buf, nbytes);
    o->len += nbytes;
    return (int64_t)nbytes;
}

int pb_dada_end_write(pb_dada *d)
{
    if (!d || d->mode != PB_DADA_WRITE) return -1;
    if (!d->in_obs) return 0;
    d->r->obs[d->r->nobs - 1].closed = 1;
    d->in_obs = 0;
    return 0;
}

void pb_dada_close(pb_dada *d) { free(d); }
