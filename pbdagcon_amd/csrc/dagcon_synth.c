/*
 * dagcon_synth.c -- deterministic synthetic pileup generator (host utility).
 *
 * Produces, per target, a random backbone and `coverage` reads derived from it
 * by i.i.d. edits (substitution / insertion runs / deletion), emitted as the
 * exact edit alignment in the layout the C-ABI batch uses (one query blob, one
 * target blob, per-alignment start/offset/length) -- the same strings a BLASR
 * -m 5 record carries in fields 16 and 18 (reference Alignment.cpp:44-80).
 * SURVEY.md section 8(d) defines the workload shapes.
 *
 * PRNG: splitmix64-seeded xoshiro256**, seed = base_seed + target_index, so a
 * target's data does not depend on how the batch is sharded.
 */
#include <stddef.h>
#include <stdint.h>

typedef struct rng { uint64_t s[4]; } rng;

static uint64_t splitmix64(uint64_t *x) {
    uint64_t z = (*x += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static void rng_seed(rng *r, uint64_t seed) {
    for (int i = 0; i < 4; i++) r->s[i] = splitmix64(&seed);
}
static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
static inline uint64_t rng_next(rng *r) {
    uint64_t *s = r->s;
    uint64_t result = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
    s[2] ^= t; s[3] = rotl(s[3], 45);
    return result;
}
/* uniform in [0,1) with 24 bits: thresholds below are compared in integers */
static inline uint32_t rng_u24(rng *r) { return (uint32_t)(rng_next(r) >> 40); }

typedef struct dagcon_synth_params {
    uint32_t tlen;        /* backbone length */
    uint32_t coverage;    /* reads per target */
    uint32_t sub_ppm;     /* substitution probability per target base, parts per million */
    uint32_t ins_open_ppm;/* probability of opening an insertion run after a column */
    uint32_t ins_ext_ppm; /* probability of extending an open insertion run */
    uint32_t del_ppm;     /* deletion probability per target base */
    uint32_t min_span_ppm;/* minimum read span as a fraction of tlen (1e6 = full length) */
    uint32_t reserved;
} dagcon_synth_params;

static const char BASES[4] = {'A', 'C', 'G', 'T'};

static inline int hit(rng *r, uint32_t ppm) {
    /* 24-bit draw against ppm scaled to 2^24 */
    return rng_u24(r) < (uint32_t)(((uint64_t)ppm << 24) / 1000000ull);
}

/*
 * Generates one target.  When qblob/tblob are NULL only sizes are computed.
 * backbone (optional, tlen bytes) receives the backbone.  starts/lens receive
 * per-read 1-based start and column count; offs are relative to blob_base.
 * Returns the total number of columns written (sum of lens).
 */
uint64_t dagcon_synth_target(const dagcon_synth_params *p, uint64_t seed,
                             char *backbone, uint32_t *starts, uint32_t *lens,
                             uint64_t *offs, uint64_t blob_base, char *qblob, char *tblob) {
    rng r;
    rng_seed(&r, seed);
    const uint32_t L = p->tlen;
    /* backbone bases are drawn first and re-drawn identically per read by
     * replaying the generator: keep them in a small rolling state instead of
     * requiring a buffer when the caller passed none. */
    rng bb_rng = r;
    for (uint32_t i = 0; i < L; i++) {
        char b = BASES[rng_next(&r) >> 62];
        if (backbone) backbone[i] = b;
    }
    uint64_t total = 0;
    for (uint32_t k = 0; k < p->coverage; k++) {
        uint32_t span = L, s0 = 0;
        if (p->min_span_ppm < 1000000u && L > 0) {
            uint32_t min_span = (uint32_t)(((uint64_t)L * p->min_span_ppm) / 1000000ull);
            if (min_span < 1) min_span = 1;
            span = min_span + (uint32_t)(rng_next(&r) % (uint64_t)(L - min_span + 1));
            s0 = (uint32_t)(rng_next(&r) % (uint64_t)(L - span + 1));
        }
        /* position a replay of the backbone stream at s0 */
        rng bb = bb_rng;
        for (uint32_t i = 0; i < s0; i++) rng_next(&bb);
        uint64_t n = 0;
        char *q = qblob ? qblob + total : NULL, *t = tblob ? tblob + total : NULL;
        for (uint32_t i = 0; i < span; i++) {
            char b = BASES[rng_next(&bb) >> 62];
            if (hit(&r, p->del_ppm)) {
                if (q) { q[n] = '-'; t[n] = b; }
                n++;
            } else if (hit(&r, p->sub_ppm)) {
                char x = BASES[((b == 'A' ? 0 : b == 'C' ? 1 : b == 'G' ? 2 : 3) + 1 + (rng_next(&r) % 3)) & 3];
                if (q) { q[n] = x; t[n] = b; }
                n++;
            } else {
                if (q) { q[n] = b; t[n] = b; }
                n++;
            }
            if (hit(&r, p->ins_open_ppm)) {
                do {
                    char x = BASES[rng_next(&r) >> 62];
                    if (q) { q[n] = x; t[n] = '-'; }
                    n++;
                } while (hit(&r, p->ins_ext_ppm));
            }
        }
        if (starts) starts[k] = s0 + 1;
        if (lens) lens[k] = (uint32_t)n;
        if (offs) offs[k] = blob_base + total;
        total += n;
    }
    return total;
}
