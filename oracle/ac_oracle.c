/*
 * oracle/ac_oracle.c -- TEST INFRASTRUCTURE ONLY (see ac_oracle.h).
 *
 * Restates, in plain C, the algorithm that the reference's hot path runs:
 *
 *   reference call site                         restated here
 *   -----------------------------------------   ---------------------------
 *   AhoCorasickBuilder..build  src/lib.rs:186-215, 401-406   aco_build()
 *   try_find_iter              src/lib.rs:59                 aco_find_iter()
 *   try_find_overlapping_iter  src/lib.rs:53                 aco_find_overlapping_iter()
 *   eager MatchError->ValueError src/lib.rs:36-39,52-54      return -1
 *   get_byte_to_code_point     src/lib.rs:73-88              aco_byte_to_code_point()
 *   (pid, b2c[start], b2c[end]) src/lib.rs:240-246           aco_find_str()
 *
 * The arithmetic lives in crate `aho-corasick` 1.1.4 (Cargo.lock:6-7), absent
 * from /root/reference; its published algorithm is restated function by
 * function (names in comments are the crate's):
 *   nfa::noncontiguous::Compiler::{build_trie, fill_failure_transitions,
 *   add_unanchored_start_state_loop, add_dead_state_loop}, NFA::{add_match,
 *   copy_matches, next_state}, util::alphabet::ByteClassSet,
 *   dfa::Builder::build_from_noncontiguous (dense, stride = next_pow2(classes),
 *   premultiplied ids, special states low), automaton::{try_find_fwd,
 *   try_find_overlapping_fwd}, FindIter / FindOverlappingIter.
 * Prefilters (memchr/Teddy) never change results and are omitted.
 *
 * This is deliberately a DIFFERENT algorithm from the product's device path
 * (which enumerates all occurrences with a Standard automaton and resolves the
 * match kind afterwards), so agreement between the two is meaningful.
 */
#include "ac_oracle.h"
#include <stdlib.h>
#include <string.h>

#define DEAD 0u
#define FAIL 1u
#define START 2u

typedef struct {
    uint32_t sparse;  /* head of byte-sorted linked list in tr[] (0 = none) */
    uint32_t matches; /* head of match list in mp[] (0 = none)             */
    uint32_t mtail;   /* tail of match list                                */
    uint32_t fail;
    uint32_t depth;
} nstate;
typedef struct { uint32_t next; uint32_t link; uint8_t byte; } ntrans;
typedef struct { uint32_t pid; uint32_t link; } nmatch;

struct aco {
    int match_kind, kind;
    uint64_t npat;
    uint32_t *plen;
    uint64_t minlen, maxlen;
    /* noncontiguous NFA */
    uint32_t ns, ns_cap;
    nstate *st;
    ntrans *tr; uint32_t ntr, tr_cap;
    nmatch *mp; uint32_t nmp, mp_cap;
    uint32_t start_dense[256]; /* transitions of the unanchored start state */
    /* dense DFA */
    uint8_t classes[256];
    uint32_t nclasses, stride, stride2;
    uint32_t nd;
    uint32_t *dtrans;      /* nd * stride, premultiplied targets */
    uint32_t d_start;      /* premultiplied */
    uint32_t d_max_match;  /* premultiplied id of last match state; DEAD = 0 */
    uint32_t *d_moff;      /* nd + 1 */
    uint32_t *d_mpid;
};

/* ---------- small helpers ---------- */
static int grow(void **p, uint32_t *cap, uint32_t need, size_t esz) {
    if (need <= *cap) return 0;
    uint32_t nc = *cap ? *cap : 1024;
    while (nc < need) nc = nc + nc / 2 + 16;
    void *q = realloc(*p, (size_t)nc * esz);
    if (!q) return -1;
    *p = q; *cap = nc;
    return 0;
}

static uint32_t alloc_state(aco_t *a, uint32_t depth) {
    if (grow((void **)&a->st, &a->ns_cap, a->ns + 1, sizeof(nstate))) return 0xFFFFFFFFu;
    nstate *s = &a->st[a->ns];
    s->sparse = 0; s->matches = 0; s->mtail = 0; s->depth = depth;
    s->fail = START; /* crate: alloc_state sets fail = start_unanchored_id */
    return a->ns++;
}

/* NFA::follow_transition */
static inline uint32_t follow(const aco_t *a, uint32_t sid, uint8_t b) {
    if (sid == START) return a->start_dense[b];
    if (sid == DEAD) return DEAD; /* add_dead_state_loop */
    for (uint32_t l = a->st[sid].sparse; l; l = a->tr[l].link) {
        if (a->tr[l].byte == b) return a->tr[l].next;
        if (a->tr[l].byte > b) break;
    }
    return FAIL;
}

/* NFA::add_transition (sorted insert) */
static int add_transition(aco_t *a, uint32_t prev, uint8_t b, uint32_t next) {
    if (prev == START) { a->start_dense[b] = next; }
    if (grow((void **)&a->tr, &a->tr_cap, a->ntr + 1, sizeof(ntrans))) return -1;
    uint32_t id = a->ntr++;
    a->tr[id].byte = b; a->tr[id].next = next;
    uint32_t *pl = &a->st[prev].sparse;
    while (*pl && a->tr[*pl].byte < b) pl = &a->tr[*pl].link;
    a->tr[id].link = *pl;
    *pl = id;
    return 0;
}

/* NFA::add_match: append at the tail (duplicates therefore stay in id order) */
static int add_match(aco_t *a, uint32_t sid, uint32_t pid) {
    if (grow((void **)&a->mp, &a->mp_cap, a->nmp + 1, sizeof(nmatch))) return -1;
    uint32_t id = a->nmp++;
    a->mp[id].pid = pid; a->mp[id].link = 0;
    if (a->st[sid].mtail) a->mp[a->st[sid].mtail].link = id;
    else a->st[sid].matches = id;
    a->st[sid].mtail = id;
    return 0;
}

/* NFA::copy_matches(src, dst): append src's list to dst's tail */
static int copy_matches(aco_t *a, uint32_t src, uint32_t dst) {
    for (uint32_t l = a->st[src].matches; l; l = a->mp[l].link)
        if (add_match(a, dst, a->mp[l].pid)) return -1;
    return 0;
}

/* NFA::next_state (unanchored): follow fail links until a transition exists */
static inline uint32_t nfa_next(const aco_t *a, uint32_t sid, uint8_t b) {
    for (;;) {
        uint32_t n = follow(a, sid, b);
        if (n != FAIL) return n;
        sid = a->st[sid].fail;
    }
}

/* ---------- construction ---------- */
static int build_trie(aco_t *a, const uint8_t *blob, const uint64_t *off) {
    for (uint64_t i = 0; i < a->npat; i++) {
        const uint8_t *pat = blob + off[i];
        uint64_t len = off[i + 1] - off[i];
        a->plen[i] = (uint32_t)len;
        if (len < a->minlen) a->minlen = len;
        if (len > a->maxlen) a->maxlen = len;
        uint32_t prev = START;
        int saw_match = 0, skipped = 0;
        for (uint64_t d = 0; d < len; d++) {
            uint8_t b = pat[d];
            /* leftmost-first: a previously added pattern that is a prefix of
             * this one makes this one unmatchable; stop adding it. */
            saw_match = saw_match || a->st[prev].matches != 0;
            if (a->match_kind == 1 && saw_match) { skipped = 1; break; }
            uint32_t next = follow(a, prev, b);
            /* during build_trie the start state's missing transitions are
             * FAIL (init_unanchored_start_state); we keep 0 in start_dense
             * for "absent" until the self-loop is added below. */
            if (prev == START && next == 0) next = FAIL;
            if (next != FAIL) {
                prev = next;
            } else {
                uint32_t ns = alloc_state(a, (uint32_t)d + 1);
                if (ns == 0xFFFFFFFFu) return -1;
                if (add_transition(a, prev, b, ns)) return -1;
                prev = ns;
            }
        }
        if (!skipped)
            if (add_match(a, prev, (uint32_t)i)) return -1;
    }
    return 0;
}

static int fill_failure_transitions(aco_t *a) {
    int is_leftmost = a->match_kind != 0;
    uint32_t *queue = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)a->ns);
    if (!queue) return -1;
    uint32_t qh = 0, qt = 0;
    /* all non-self transitions out of the start state */
    for (uint32_t l = a->st[START].sparse; l; l = a->tr[l].link) {
        uint32_t nx = a->tr[l].next;
        if (nx == START) continue;
        queue[qt++] = nx;
        if (is_leftmost && a->st[nx].matches) a->st[nx].fail = DEAD;
    }
    while (qh < qt) {
        uint32_t id = queue[qh++];
        for (uint32_t l = a->st[id].sparse; l; l = a->tr[l].link) {
            uint8_t b = a->tr[l].byte;
            uint32_t nx = a->tr[l].next;
            queue[qt++] = nx; /* trie: every state has exactly one parent */
            if (is_leftmost && a->st[nx].matches) {
                a->st[nx].fail = DEAD;
                continue;
            }
            uint32_t f = a->st[id].fail;
            while (follow(a, f, b) == FAIL) f = a->st[f].fail;
            f = follow(a, f, b);
            a->st[nx].fail = f;
            if (copy_matches(a, f, nx)) { free(queue); return -1; }
        }
        /* (start state is never a match state: empty patterns are rejected
         * upstream, so the crate's copy_matches(start, id) is a no-op) */
    }
    free(queue);
    return 0;
}

static void byte_classes(aco_t *a, const uint8_t *blob, const uint64_t *off) {
    /* util::alphabet::ByteClassSet: set_range(b, b) for every pattern byte */
    uint8_t bits[256];
    memset(bits, 0, sizeof bits);
    uint64_t total = off[a->npat];
    for (uint64_t i = 0; i < total; i++) {
        uint8_t b = blob[i];
        if (b > 0) bits[b - 1] = 1;
        bits[b] = 1;
    }
    uint32_t cls = 0;
    for (int b = 0; b < 256; b++) {
        a->classes[b] = (uint8_t)cls;
        if (bits[b] && b != 255) cls++;
    }
    a->nclasses = (uint32_t)a->classes[255] + 1;
    uint32_t s = 1, s2 = 0;
    while (s < a->nclasses) { s <<= 1; s2++; }
    a->stride = s; a->stride2 = s2;
}

static int build_dfa(aco_t *a) {
    /* order: DEAD, match states, START, everything else (special states
     * low, so that `sid <= d_max_match` identifies dead-or-match). */
    uint32_t ns = a->ns;
    uint32_t *map = (uint32_t *)malloc(sizeof(uint32_t) * ns); /* nfa -> dfa index */
    uint32_t *order = (uint32_t *)malloc(sizeof(uint32_t) * ns);
    uint32_t *bfs = (uint32_t *)malloc(sizeof(uint32_t) * ns);
    if (!map || !order || !bfs) return -1;
    uint32_t nd = 0;
    map[DEAD] = nd; order[nd++] = DEAD;
    for (uint32_t s = START + 1; s < ns; s++)
        if (a->st[s].matches) { map[s] = nd; order[nd++] = s; }
    uint32_t last_match = nd - 1;
    map[START] = nd; order[nd++] = START;
    for (uint32_t s = START + 1; s < ns; s++)
        if (!a->st[s].matches) { map[s] = nd; order[nd++] = s; }
    map[FAIL] = 0xFFFFFFFFu;
    a->nd = nd;
    a->d_max_match = last_match << a->stride2;
    a->d_start = map[START] << a->stride2;
    a->dtrans = (uint32_t *)calloc((size_t)nd * a->stride, sizeof(uint32_t));
    if (!a->dtrans) return -1;
    /* representative byte per class */
    uint8_t rep[256];
    for (int b = 255; b >= 0; b--) rep[a->classes[b]] = (uint8_t)b;
    /* BFS order so that fail(s)'s row is finished before s's row */
    uint32_t qh = 0, qt = 0;
    bfs[qt++] = START;
    while (qh < qt) {
        uint32_t id = bfs[qh++];
        for (uint32_t l = a->st[id].sparse; l; l = a->tr[l].link)
            if (a->tr[l].next != START) bfs[qt++] = a->tr[l].next;
    }
    /* DEAD row: all zeros (DEAD = 0) already */
    for (uint32_t k = 0; k < qt; k++) {
        uint32_t s = bfs[k];
        uint32_t *row = a->dtrans + (size_t)map[s] * a->stride;
        if (s == START) {
            for (uint32_t c = 0; c < a->nclasses; c++)
                row[c] = map[a->start_dense[rep[c]]] << a->stride2;
        } else {
            uint32_t f = a->st[s].fail;
            const uint32_t *frow = a->dtrans + (size_t)map[f] * a->stride;
            memcpy(row, frow, sizeof(uint32_t) * a->stride);
            for (uint32_t l = a->st[s].sparse; l; l = a->tr[l].link)
                row[a->classes[a->tr[l].byte]] = map[a->tr[l].next] << a->stride2;
        }
    }
    /* match lists (DFA::set_matches) */
    a->d_moff = (uint32_t *)calloc((size_t)nd + 1, sizeof(uint32_t));
    if (!a->d_moff) return -1;
    for (uint32_t i = 0; i < nd; i++) {
        uint32_t c = 0;
        for (uint32_t l = a->st[order[i]].matches; l; l = a->mp[l].link) c++;
        a->d_moff[i + 1] = a->d_moff[i] + c;
    }
    a->d_mpid = (uint32_t *)malloc(sizeof(uint32_t) * ((size_t)a->d_moff[nd] + 1));
    if (!a->d_mpid) return -1;
    for (uint32_t i = 0; i < nd; i++) {
        uint32_t k = a->d_moff[i];
        for (uint32_t l = a->st[order[i]].matches; l; l = a->mp[l].link)
            a->d_mpid[k++] = a->mp[l].pid;
    }
    free(map); free(order); free(bfs);
    return 0;
}

aco_t *aco_build(const uint8_t *blob, const uint64_t *off, uint64_t n,
                 int match_kind, int kind) {
    for (uint64_t i = 0; i < n; i++)
        if (off[i + 1] == off[i]) return NULL; /* empty pattern */
    aco_t *a = (aco_t *)calloc(1, sizeof(aco_t));
    if (!a) return NULL;
    a->match_kind = match_kind; a->kind = kind; a->npat = n;
    a->minlen = UINT64_MAX; a->maxlen = 0;
    a->plen = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(n + 1));
    a->ntr = 1; a->nmp = 1; /* index 0 = "none" */
    if (grow((void **)&a->tr, &a->tr_cap, 16, sizeof(ntrans))) goto bad;
    if (grow((void **)&a->mp, &a->mp_cap, 16, sizeof(nmatch))) goto bad;
    /* DEAD, FAIL, START */
    alloc_state(a, 0); alloc_state(a, 0); alloc_state(a, 0);
    a->st[DEAD].fail = DEAD;
    memset(a->start_dense, 0, sizeof a->start_dense); /* 0 == absent for now */
    if (build_trie(a, blob, off)) goto bad;
    /* add_unanchored_start_state_loop */
    for (int b = 0; b < 256; b++)
        if (a->start_dense[b] == 0) a->start_dense[b] = START;
    if (fill_failure_transitions(a)) goto bad;
    byte_classes(a, blob, off);
    if (kind == 2)
        if (build_dfa(a)) goto bad;
    if (n == 0) { a->minlen = 0; }
    return a;
bad:
    aco_free(a);
    return NULL;
}

void aco_free(aco_t *a) {
    if (!a) return;
    free(a->plen); free(a->st); free(a->tr); free(a->mp);
    free(a->dtrans); free(a->d_moff); free(a->d_mpid);
    free(a);
}

uint64_t aco_num_states(const aco_t *a) { return a->ns; }
uint64_t aco_num_classes(const aco_t *a) { return a->nclasses; }
uint64_t aco_max_pattern_len(const aco_t *a) { return a->maxlen; }
uint64_t aco_min_pattern_len(const aco_t *a) { return a->minlen; }

/* ---------- search ---------- */
typedef struct { uint64_t pid, start, end; int found; } omatch;

/* automaton::try_find_fwd on the noncontiguous NFA */
static omatch try_find_fwd_nfa(const aco_t *a, const uint8_t *hay,
                               uint64_t at, uint64_t end) {
    omatch mat = {0, 0, 0, 0};
    int earliest = a->match_kind == 0;
    uint32_t sid = START;
    while (at < end) {
        sid = nfa_next(a, sid, hay[at]);
        if (sid == DEAD) return mat;
        uint32_t ml = a->st[sid].matches;
        if (ml) {
            uint32_t pid = a->mp[ml].pid; /* match_pattern(sid, 0) */
            mat.pid = pid; mat.end = at + 1; mat.start = at + 1 - a->plen[pid];
            mat.found = 1;
            if (earliest) return mat;
        }
        at++;
    }
    return mat;
}

/* automaton::try_find_fwd on the dense DFA: the crate's inner loop shape
 * (class map, one dependent u32 load per byte, special-state range check) */
static omatch try_find_fwd_dfa(const aco_t *a, const uint8_t *hay,
                               uint64_t at, uint64_t end) {
    omatch mat = {0, 0, 0, 0};
    int earliest = a->match_kind == 0;
    const uint32_t *trans = a->dtrans;
    const uint8_t *cls = a->classes;
    const uint32_t max_special = a->d_max_match;
    uint32_t sid = a->d_start;
    while (at < end) {
        sid = trans[sid + cls[hay[at]]];
        if (sid <= max_special) {
            if (sid == DEAD) return mat;
            uint32_t pid = a->d_mpid[a->d_moff[sid >> a->stride2]];
            mat.pid = pid; mat.end = at + 1; mat.start = at + 1 - a->plen[pid];
            mat.found = 1;
            if (earliest) return mat;
        }
        at++;
    }
    return mat;
}

/* FindIter: repeated try_find with span.start = previous match end */
int64_t aco_find_iter(const aco_t *a, const uint8_t *hay, uint64_t len,
                      uint64_t *out, uint64_t cap) {
    uint64_t at = 0; int64_t n = 0;
    while (at <= len) {
        omatch m = a->kind == 2 ? try_find_fwd_dfa(a, hay, at, len)
                                : try_find_fwd_nfa(a, hay, at, len);
        if (!m.found) break;
        if ((uint64_t)n < cap && out) {
            out[3 * n] = m.pid; out[3 * n + 1] = m.start; out[3 * n + 2] = m.end;
        }
        n++;
        at = m.end; /* patterns are non-empty, so end > start >= at */
    }
    return n;
}

int64_t aco_count_iter(const aco_t *a, const uint8_t *hay, uint64_t len) {
    return aco_find_iter(a, hay, len, NULL, 0);
}

/* FindOverlappingIter / try_find_overlapping_fwd: state persists; at a match
 * state every pattern of its list is yielded in list order. */
int64_t aco_find_overlapping_iter(const aco_t *a, const uint8_t *hay,
                                  uint64_t len, uint64_t *out, uint64_t cap) {
    if (a->match_kind != 0) return -1; /* MatchError::UnsupportedOverlapping */
    int64_t n = 0;
    if (a->kind == 2) {
        uint32_t sid = a->d_start;
        for (uint64_t at = 0; at < len; at++) {
            sid = a->dtrans[sid + a->classes[hay[at]]];
            if (sid <= a->d_max_match) {
                if (sid == DEAD) break;
                uint32_t si = sid >> a->stride2;
                for (uint32_t k = a->d_moff[si]; k < a->d_moff[si + 1]; k++) {
                    uint32_t pid = a->d_mpid[k];
                    if ((uint64_t)n < cap && out) {
                        out[3 * n] = pid; out[3 * n + 1] = at + 1 - a->plen[pid];
                        out[3 * n + 2] = at + 1;
                    }
                    n++;
                }
            }
        }
    } else {
        uint32_t sid = START;
        for (uint64_t at = 0; at < len; at++) {
            sid = nfa_next(a, sid, hay[at]);
            if (sid == DEAD) break;
            for (uint32_t l = a->st[sid].matches; l; l = a->mp[l].link) {
                uint32_t pid = a->mp[l].pid;
                if ((uint64_t)n < cap && out) {
                    out[3 * n] = pid; out[3 * n + 1] = at + 1 - a->plen[pid];
                    out[3 * n + 2] = at + 1;
                }
                n++;
            }
        }
    }
    return n;
}

/* src/lib.rs:73-88 */
void aco_byte_to_code_point(const uint8_t *hay, uint64_t len, uint64_t *out) {
    for (uint64_t i = 0; i <= len; i++) out[i] = UINT64_MAX;
    uint64_t cp = 0, max_cp = 0;
    for (uint64_t i = 0; i < len; i++) {
        if ((hay[i] & 0xC0) != 0x80) { /* char_indices(): a char starts here */
            out[i] = cp; max_cp = cp; cp++;
        }
    }
    if (len) out[len] = max_cp + 1;
}

/* src/lib.rs:229-249 */
int64_t aco_find_str(const aco_t *a, const uint8_t *utf8, uint64_t len,
                     int overlapping, uint64_t *out, uint64_t cap) {
    uint64_t *b2c = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(len + 1));
    if (!b2c) return -2;
    aco_byte_to_code_point(utf8, len, b2c);
    int64_t n = overlapping ? aco_find_overlapping_iter(a, utf8, len, out, cap)
                            : aco_find_iter(a, utf8, len, out, cap);
    if (n > 0 && out) {
        uint64_t m = (uint64_t)n < cap ? (uint64_t)n : cap;
        for (uint64_t i = 0; i < m; i++) {
            out[3 * i + 1] = b2c[out[3 * i + 1]];
            out[3 * i + 2] = b2c[out[3 * i + 2]];
        }
    }
    free(b2c);
    return n;
}
