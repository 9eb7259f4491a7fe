// resco_step.h -- the fused MultiSignal.step() of one environment: FSM + step_length one-second ticks of the
// microsimulation + Signal.observe + states / rewards, as a sequence of PHASES over the vehicle slots of the environment.
//
// The body is written against a small execution interface (`Exec`):
//   ex.phase(id, f)         runs f(tid) for every thread of the workgroup and ends with a workgroup barrier (id: 0..15 names
//                           the phase for the optional per-phase timers: 0-2 load, 4 plan, 5 check, 6 move, 11-14 observe)
//   ex.role_begin/role_end  optional timers of what one WAVE does inside a phase (ids 3, 7-10, 15)
//   ex.B                    threads per workgroup (a multiple of 64)
//   ex.wave_of(tid)         tid / 64, known to be the same for the 64 threads of a wave (a scalar on the GPU)
//   ex.next_chunk(c, w, i, n)  the next work chunk of wave w (its i-th call in this phase, n waves): on the GPU ONE atomic on the LDS
//                           counter c per wave (dynamic: whichever wave is free takes the next chunk), on the host i * n + w --
//                           any assignment of chunks to waves gives the same result, a phase being order independent
//   ex.plan_long / ex.lc_decide / ex.move_long   the long code paths of a tick for ONE slot (ExecInline below: inlined; -DRS_CALL_LONG: called as functions
//                           with a register allocation of their own -- measured, not adopted)
// On the GPU (resco_sim.hip) one workgroup = one environment; the state lives in LDS for the whole env-step and phase() is
// `f(threadIdx.x); __syncthreads()`.  The CPU tests compile the very same source for the host (tests/hostemu), where
// phase() calls f for tid = 0 .. B-1 in turn (in any order: a phase never reads what another thread writes in the same
// phase, except through the order-independent atomics below).
//
// Including file provides: RS_DEV / RS_HD / RS_MEM / RS_CARVE (function qualifiers), RS_G (marks a pointer as global
// memory), RS_SMEM (base of the working memory, see LPtr), the atomics rs_atomic_min / max / add / or / and / cas /
// fetch_add / fetch_or on 32-bit words, the wave-level helpers rs_wave_add / rs_wave_max / rs_wave_ticket, the bit helpers
// rs_popc / rs_ffs / rs_ffsll / rs_clzll, rs_f2h (float -> half bits), rs_int_as_float / rs_float_as_int and <math.h>.
#pragma once
#include "resco_tables.h"

// ------------------------------------------------------------------------------------------------ HBM layout
// Kernel arguments are kept SMALL on purpose: every pointer passed by value costs two SGPRs for the whole kernel.  The
// per-slot state and the outputs are therefore ONE allocation each, with field addresses computed from (base, N*C).
struct State {      // env-major SoA in HBM: field[env][slot]
    char *base;
    size_t nc;          // N * C
    int32_t *trip_log;  // [N][n_trips][4] or NULL
    int32_t *env;       // [N][4] t, n_inserted, hw, n_active
    int32_t *tls;       // [N][S][TLS_W] phase, left, next_phase, departures since the last observe
    long long *stats;   // [N][10]
    uint16_t *dep_next; // [N][n_dep] head of every departure lane's backlog (TRIP_NONE: exhausted)
    uint32_t *mail;     // [N][ceil(C/32)] bit per slot: a cooperation request of the last tick waits in its mailboxes (see SFQ_MAIL)
    RS_HD float *pos() const { return (float *)base; }
    RS_HD float *speed() const { return RS_G((float *)(base + 4 * nc)); }
    RS_HD float *accel() const { return RS_G((float *)(base + 8 * nc)); }
    RS_HD float *tloss() const { return RS_G((float *)(base + 12 * nc)); }
    RS_HD float *sf() const { return RS_G((float *)(base + 16 * nc)); }
    // cooperation requests, double-buffered by tick parity: the lane-change decisions of tick t write buffer t & 1 while the
    // plans of the same phase read (and clear) the requests of tick t - 1 in the other one
    RS_HD uint32_t *coop(int par) const { return RS_G((uint32_t *)(base + (par ? 43 : 20) * nc)); }
    RS_HD uint32_t *cooplead(int par) const { return RS_G((uint32_t *)(base + (par ? 47 : 24) * nc)); }      // (the same)
    RS_HD uint16_t *lane() const { return RS_G((uint16_t *)(base + 28 * nc)); }
    RS_HD uint16_t *trip() const { return RS_G((uint16_t *)(base + 30 * nc)); }
    RS_HD uint16_t *cursor() const { return RS_G((uint16_t *)(base + 32 * nc)); }
    RS_HD uint16_t *swait() const { return RS_G((uint16_t *)(base + 34 * nc)); }
    RS_HD uint16_t *rwait() const { return RS_G((uint16_t *)(base + 36 * nc)); }
    RS_HD uint16_t *depart() const { return RS_G((uint16_t *)(base + 38 * nc)); }
    RS_HD uint16_t *wtot() const { return RS_G((uint16_t *)(base + 40 * nc)); }
    RS_HD uint8_t *owner() const { return RS_G((uint8_t *)(base + 42 * nc)); }
    static size_t bytes(size_t nc_) { return 51 * nc_; }
};

struct Out {        // one allocation; n = N, o = n_obs, s = n_signals, lm = lanes of the largest signal
    char *base;
    int32_t n, o, s, lm;
    RS_HD size_t a5() const { return (size_t)n * o * 5 * 4; }      // one [N][n_obs][5] f32 block
    RS_HD size_t ns() const { return (size_t)n * s * 4; }          // one [N][S] 4-byte block
    RS_HD float *lane_agg() const { return (float *)base; }
    RS_HD float *drq_norm() const { return RS_G((float *)(base + a5())); }
    RS_HD float *wait() const { return RS_G((float *)(base + 2 * a5())); }
    RS_HD float *wait_norm() const { return RS_G((float *)(base + 2 * a5() + ns())); }
    RS_HD int32_t *phase() const { return RS_G((int32_t *)(base + 2 * a5() + 2 * ns())); }
    RS_HD int32_t *pressure() const { return RS_G((int32_t *)(base + 2 * a5() + 3 * ns())); }
    RS_HD int32_t *queue_sum() const { return RS_G((int32_t *)(base + 2 * a5() + 4 * ns())); }
    RS_HD int32_t *queue_max() const { return RS_G((int32_t *)(base + 2 * a5() + 5 * ns())); }
    RS_HD int32_t *mplight() const { return RS_G((int32_t *)(base + 2 * a5() + 6 * ns())); }
    RS_HD int32_t *wave() const { return RS_G((int32_t *)(base + 2 * a5() + 19 * ns())); }
    RS_HD int32_t *arrivals() const { return RS_G((int32_t *)(base + 2 * a5() + 31 * ns())); }      // [N][S] |Signal.arrivals| of the last observe
    RS_HD int32_t *departures() const { return RS_G((int32_t *)(base + 2 * a5() + 32 * ns())); }    // [N][S] |Signal.departures|
    RS_HD float *mplight_full() const { return RS_G((float *)(base + 2 * a5() + 33 * ns())); }      // [N][S][49]
    RS_HD int32_t *lane_arr() const { return RS_G((int32_t *)(base + 2 * a5() + 82 * ns())); }     // [N][n_obs] vehicles of the lane that are in their signal's `arrivals` set
    RS_HD uint16_t *drq_f16() const { return RS_G((uint16_t *)(base + 2 * a5() + 82 * ns() + a5() / 5)); }
    RS_HD size_t bytes() const { return 2 * a5() + 82 * ns() + a5() / 5 + (size_t)n * s * lm * 5 * 2 + 64; }
};

// ------------------------------------------------------------------------------------------------ device math
RS_DEV uint32_t rs_rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
RS_DEV uint32_t d_hash(uint32_t seed, uint32_t env, uint32_t trip, uint32_t tick, uint32_t stream) {
    uint32_t h = seed;
    const uint32_t w[4] = {env, trip, tick, stream};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint32_t k = w[i];
        k *= 0xcc9e2d51u; k = rs_rotl32(k, 15); k *= 0x1b873593u;
        h ^= k; h = rs_rotl32(h, 13); h = h * 5u + 0xe6546b64u;
    }
    h ^= 16u;
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}
RS_DEV float d_u01(uint32_t h) { return (float)(h >> 8) * (1.0f / 16777216.0f); }

// a / b, correctly rounded.  The including file may define RS_DIV: the device build uses the hardware's own Newton sequence (v_rcp_f32 + seven
// fma / mul, what the compiler emits for `/`) WITHOUT the range scaling around it (v_div_scale x 2, v_div_fmas, v_div_fixup and the wait states
// their condition codes need): every quotient of the model has a finite, non-zero denominator and operands far from the ends of the
// exponent range, where the scaled and the unscaled sequence are the same operations -- bit-identical, 3-5 issue slots less per division,
// and a plan + move execute seven of them per vehicle and tick.
#ifndef RS_DIV
#define RS_DIV(a, b) ((a) / (b))
#endif
// Krauss (SUMO MSCFModel, Euler update, dt = 1 s) [SUMO-K]
RS_DEV float d_brake_gap(float v, float b) {
    const int steps = (int)RS_DIV(v, b);
    const float fs = (float)steps;
    return fs * v - b * fs * (fs + 1.0f) * 0.5f;
}
RS_DEV float d_stop_speed(float gap, float b, float tau) {
    const float g = gap - 0.001f;
    if (g < 0.0f) return 0.0f;
    const float q = 1.0f + 4.0f * ((RS_DIV(2.0f * g, b) - tau) + tau * tau);
    const float n = floorf(0.5f - (tau + sqrtf(q) * -0.5f));
    const float h = 0.5f * n * (n - 1.0f) * b + n * b * tau;
    const float r = RS_DIV(g - h, n + tau);
    return n * b + r;
}
RS_DEV float d_free_speed(float dist, float target, float b) {
    if (dist < target) return target;
    const float t2 = b + 2.0f * target;
    float y = RS_DIV((sqrtf(t2 * t2 + 8.0f * b * dist) - b) * 0.5f - target, b);
    if (y < 0.0f) y = 0.0f;
    const float yf = floorf(y);
    const float exact = (yf * yf + yf) * 0.5f * b + yf * target + (y > yf ? target : 0.0f);
    float rest = dist - exact;
    if (rest < 0.0f) rest = 0.0f;
    return RS_DIV(rest, yf + 1.0f) + yf * b + target;
}
RS_DEV float d_follow_speed(float gap, float vl, float b, float bl, float tau) {
    const float bm = b > bl ? b : bl;
    return d_stop_speed(gap + d_brake_gap(vl, bm), b, tau);
}
// speedFactor of a trip in units of 1 / RM_SF_QUANT (oracle: speed_factor): carried in 16 bits of the vehicle's Node
RS_DEV int speed_factor_q(const KParams &P, int env, int trip, const float *vt) {
    float f = vt[VT_SF_MEAN];
    if (P.speed_dev) {
        float s = 0.0f;
#pragma unroll
        for (uint32_t i = 0; i < 4; ++i) s += d_u01(d_hash(P.seed, (uint32_t)env, (uint32_t)trip, 0xFFFFFFFFu, i));
        const float z = (s - 2.0f) * 1.7320508f;
        f = vt[VT_SF_MEAN] + vt[VT_SF_DEV] * z;
    }
    if (f < 0.2f) f = 0.2f;
    if (f > 2.0f) f = 2.0f;
    return (int)(f * RM_SF_QUANT + 0.5f);
}
// Node.sfq: bits 0..13 the speed factor (at most 2.0 * RM_SF_QUANT = 8192), bits 14 / 15 the MAIL FLAGS of the two tick parities: "a
// cooperation request written in a tick of parity p addresses this vehicle".  The lane-change decision that writes a request sets the
// flag (atomic OR on the record's last dword), the plan of the next tick reads its two mailboxes -- global memory -- only when it is set
// and clears it: 95 % of the plans issue no mailbox load at all, and the mailbox lines of an environment are no longer fetched once
// per launch for nothing ("Bytes per env-step").  Between launches the flags of the last tick live in State.mail.
#define SFQ_MASK 0x3FFFu
#define SFQ_MAIL(par) (0x4000u << (par))
RS_DEV float sf_of(int q) { return (float)(q & (int)SFQ_MASK) * (1.0f / RM_SF_QUANT); }


// ------------------------------------------------------------------------------------------------ working memory (LDS)
struct __attribute__((aligned(16))) Node {      // everything a NEIGHBOUR wants to know about a vehicle: one 16-byte read
    float pos;
    float speed;
    uint16_t trip;      // TRIP_NONE: free slot
    uint16_t nxt;       // next vehicle of the same grid cell (unordered), NIL terminated
    uint8_t vt;         // vType
    uint8_t fl;         // scheduling hints and the tick's lane-change decision, see FL_* / LCT_*
    uint16_t sfq;       // speedFactor in units of 1 / RM_SF_QUANT
};
// A wave executes every branch that ANY of its 64 lanes takes, so the rare, long code paths (junction look-ahead, lane-change
// searches, hand-over / arrival) are not left scattered over all waves: the vehicles that need them are queued in short
// lists and handled by consecutive threads, after the threads have gone through their own slots on the short path.
#define FL_H 1          // the plan must look beyond the end of the lane (set when the vehicle is moved / loaded / inserted)
#define FL_MH 2         // this tick's move leaves the lane, forward or sideways (even ticks)
#define FL_LC 4         // this tick's lane-change decision needs neighbour searches
#define FL_MH1 8        // FL_MH of odd ticks.  Two bits by tick parity: the move phase must not clear the bit it tests (a thread may
                        // look at its own slot after the list's thread has moved the vehicle); the next plan drops the old one
RS_DEV int fl_mh(int t) { return (t & 1) ? FL_MH1 : FL_MH; }
// The upper four bits of Node.fl: the lane-change decision of the tick -- a change that holds unless the vehicle leaves its
// lane forward in the same tick (LCT_LEFT / LCT_RIGHT), and / or a swap with the vehicle alongside, which holds in any case
#define LCT_LEFT 16
#define LCT_RIGHT 32
#define LCT_SWAP_LEFT 64
#define LCT_SWAP_RIGHT 128
struct __attribute__((aligned(8))) Aux {        // what only the owner reads, every tick: one 8-byte read
    uint16_t lane, rq, nlink;
    uint16_t swait;     // seconds the vehicle has been standing (SUMO's waiting time); HBM holds it between env-steps
};
// The layout (a table of offsets, computed once by the host: lds_carve) is read from the constant argument block.
// An array of the working memory is addressed as (RS_SMEM + offset): the including file defines RS_SMEM as THE shared
// array of the workgroup (so that every access is provably an LDS access: ds_* instructions with immediate offsets
// instead of flat ones through 64-bit pointers), the host emulation as a plain buffer.
template <class Tp> struct LPtr {
    uint32_t off;
    RS_MEM Tp &operator[](int i) const { return ((Tp *)(RS_SMEM + off))[i]; }
    RS_MEM operator Tp *() const { return (Tp *)(RS_SMEM + off); }
};
struct Lds {
    LPtr<Node> node;
    LPtr<Aux> aux;
    LPtr<float> vnx, vtp;
    LPtr<uint16_t> grid;        // ONE grid of cells (resco_tables.h: a cell carries the parity of the tick it is valid for): a tick
                                // reads the cells of its parity and pushes the moved vehicles with the other one
    uint32_t gstride;           // cells of the grid (padded to a multiple of 8)
    LPtr<int32_t> arr;          // link approach registers
    LPtr<uint16_t> dep, dep_t;  // head trip of every departure lane's backlog, and its departure second (0xFFFF: none)
    LPtr<uint32_t> alive, alive0, insm;     // bit per slot: occupied (now / at the beginning of the tick); bit per departure lane: inserts this tick
    LPtr<uint32_t> mailw;                   // bit per slot: the mail flags of the last tick, collected when the slab goes back (-> State.mail)
    LPtr<int32_t> agg_q, agg_a, agg_w, agg_m, agg_n;
    LPtr<uint32_t> agg_s;
    LPtr<int32_t> sig_arr, sig_dep;
    LPtr<int32_t> phase, left, nextp;
    LPtr<uint8_t> tstate;       // current link states of every signal, [S][tls_maxl]
    LPtr<int32_t> sc;           // scalars, see SC_*
    LPtr<uint16_t> ls_h, ls_lc, ls_mh;      // work lists (slots), `lcap` entries each; a list that overflows is ignored and
    uint32_t lcap;                          // the phase falls back to the flags (every thread handles its own slots in full)
    float cell_inv;                         // 1 / grid cell length of this handle's tables (PackedTables::cell_inv; set by rs_create)
};
#define SC_T 0
#define SC_NINS 1
#define SC_HW 2
#define SC_NACT 3
#define SC_HWNEW 4
#define SC_ROOM 5
#define SC_HWOUT 6
#define SC_NH 7         // entries of ls_h (may exceed lcap: overflow)
#define SC_NLC 8
#define SC_NMH 9        // two counters, by tick parity (the plan of tick t + 1 queues while nothing separates it from the move of t)
#define SC_CHUNK_P 11    // next work chunk of the plan phase / of the move phase (wave-level tickets, reset in C)
#define SC_CHUNK_M 12
#define SC_STATS 13
// entries of a long-path work list one wave takes at a time.  Measured on the MI355X (profiles/r04_ab_chunks.txt, ingolstadt21 x 4096):
// 64 entries 1.665 ms per launch -- exactly what the fixed roles of round 3 took --, 48: 1.674, 32: 1.784, 16: 2.109.  Handing the
// chunks out dynamically does not shorten the phases: they are not bound by the slowest role but by what all waves of the CU
// issue together.
#ifndef RS_LIST_CHUNK
#define RS_LIST_CHUNK 64
#endif
#ifndef RS_H_CHUNK
#define RS_H_CHUNK RS_LIST_CHUNK     // entries of the look-ahead list per chunk
#endif

RS_CARVE size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }
// entries of a work list for capacity C
RS_CARVE int lds_list_cap(int C) {
#ifdef RS_LIST_CAP                  // (tests: lists so short that they overflow all the time)
    (void)C;
    return RS_LIST_CAP;
#else
    return ((C / 4 + 63) / 64) * 64;
#endif
}
// Layout of the working memory: a table of offsets computed once by the host.  The arrays whose SIZE follows from the slot capacity
// alone come first, so that their OFFSETS do too: a kernel instantiated for a capacity addresses them with literals (LdsFix below)
// instead of a scalar load from the constant argument block per use -- +2 % when every offset is a literal
// (profiles/r05_ab_const_layout.txt).  The others are read from the table where they are used (an offset costs a scalar load there
// and no register in between).
RS_CARVE size_t lds_carve(Lds *L, int C, int n_cells, int n_arr, int n_dep, int n_obs, int S, int n_vt, int tls_maxl) {
    size_t o = 0;
#define CARVE(field, bytes) { if (L) L->field.off = (uint32_t)o; o += align16(bytes); }
    // ---- sizes that follow from C (LdsFix<C> restates these offsets; lds_fix_matches() compares)
    CARVE(node, (size_t)C * 16) CARVE(aux, (size_t)C * 8) CARVE(vnx, (size_t)C * 4)
    CARVE(sc, (size_t)(SC_STATS + ST_N) * 4)
    CARVE(alive, (size_t)((C + 31) / 32) * 4) CARVE(alive0, (size_t)((C + 31) / 32) * 4)
    {
        const int cap = lds_list_cap(C);
        if (L) L->lcap = (uint32_t)cap;
        CARVE(ls_h, (size_t)cap * 2) CARVE(ls_lc, (size_t)cap * 2) CARVE(ls_mh, (size_t)cap * 2)
    }
    // ---- the first array of scenario-dependent size still starts at a known place
    const uint32_t grid_off = (uint32_t)o;
    const size_t grid_bytes = align16((size_t)((n_cells + 8 + 7) & ~7) * 2);
    if (L) L->gstride = (uint32_t)((n_cells + 8 + 7) & ~7);
    CARVE(grid, (size_t)((n_cells + 8 + 7) & ~7) * 2)
    // ---- scenario-dependent
    CARVE(insm, (size_t)((n_dep + 31) / 32) * 4)
    CARVE(mailw, (size_t)((C + 31) / 32) * 4)
    CARVE(vtp, (size_t)n_vt * VT_COLS * 4)
    CARVE(phase, (size_t)S * 4) CARVE(left, (size_t)S * 4) CARVE(nextp, (size_t)S * 4)
    CARVE(sig_arr, (size_t)S * 4) CARVE(sig_dep, (size_t)S * 4)
    CARVE(tstate, (size_t)S * tls_maxl)
    CARVE(arr, (size_t)n_arr * 4) CARVE(dep, (size_t)n_dep * 2) CARVE(dep_t, (size_t)n_dep * 2)
    {
        // the per-lane aggregates of the observe phase live where the grid was (dead after the last tick), else where the next
        // speeds were, else in a region of their own
        const size_t ab = align16((size_t)n_obs * 4);
        uint32_t p = grid_off;
        if (6 * ab > grid_bytes) {
            p = (uint32_t)(align16((size_t)C * 16) + align16((size_t)C * 8));           // vnx (the third array above)
            if (6 * ab > align16((size_t)C * 4)) { p = (uint32_t)o; o += 6 * ab; }
        }
        if (L) {
            L->agg_q.off = p; L->agg_a.off = p + (uint32_t)ab; L->agg_w.off = p + (uint32_t)(2 * ab);
            L->agg_m.off = p + (uint32_t)(3 * ab); L->agg_s.off = p + (uint32_t)(4 * ab); L->agg_n.off = p + (uint32_t)(5 * ab);
        }
    }
#undef CARVE
    return o;
}

// The same working memory seen by a kernel that knows the capacity C at compile time: literals for the offsets that follow from C,
// references into the host's table (constant argument block) for the rest.  The phases are written against `L.<array>[i]` and take
// either this view or the table itself (capacity only known at run time).
template <class Tp, uint32_t OFF> struct CPtr {
    RS_MEM Tp &operator[](int i) const { return ((Tp *)(RS_SMEM + OFF))[i]; }
    RS_MEM operator Tp *() const { return (Tp *)(RS_SMEM + OFF); }
};
template <int C> struct LdsFix {
    static constexpr uint32_t a16(uint32_t x) { return (x + 15u) & ~15u; }
#ifdef RS_LIST_CAP
    static constexpr uint32_t kcap = RS_LIST_CAP;
#else
    static constexpr uint32_t kcap = ((C / 4 + 63) / 64) * 64;
#endif
    static constexpr uint32_t o_node = 0, o_aux = o_node + a16(C * 16), o_vnx = o_aux + a16(C * 8), o_sc = o_vnx + a16(C * 4),
                              o_alive = o_sc + a16((SC_STATS + ST_N) * 4), o_alive0 = o_alive + a16(((C + 31) / 32) * 4),
                              o_ls_h = o_alive0 + a16(((C + 31) / 32) * 4), o_ls_lc = o_ls_h + a16(kcap * 2), o_ls_mh = o_ls_lc + a16(kcap * 2),
                              o_grid = o_ls_mh + a16(kcap * 2);
    CPtr<Node, o_node> node;
    CPtr<Aux, o_aux> aux;
    CPtr<float, o_vnx> vnx;
    CPtr<int32_t, o_sc> sc;
    CPtr<uint32_t, o_alive> alive;
    CPtr<uint32_t, o_alive0> alive0;
    CPtr<uint16_t, o_ls_h> ls_h;
    CPtr<uint16_t, o_ls_lc> ls_lc;
    CPtr<uint16_t, o_ls_mh> ls_mh;
    CPtr<uint16_t, o_grid> grid;
    static constexpr uint32_t lcap = kcap;
    const LPtr<float> &vtp;
    const uint32_t &gstride;
    const LPtr<int32_t> &arr;
    const LPtr<uint16_t> &dep, &dep_t;
    const LPtr<uint32_t> &insm, &mailw;
    const LPtr<int32_t> &agg_q, &agg_a, &agg_w, &agg_m, &agg_n;
    const LPtr<uint32_t> &agg_s;
    const LPtr<int32_t> &sig_arr, &sig_dep, &phase, &left, &nextp;
    const LPtr<uint8_t> &tstate;
    const float &cell_inv;
    RS_MEM explicit LdsFix(const Lds &T)
        : vtp(T.vtp), gstride(T.gstride), arr(T.arr), dep(T.dep), dep_t(T.dep_t), insm(T.insm), mailw(T.mailw), agg_q(T.agg_q), agg_a(T.agg_a), agg_w(T.agg_w),
          agg_m(T.agg_m), agg_n(T.agg_n), agg_s(T.agg_s), sig_arr(T.sig_arr), sig_dep(T.sig_dep), phase(T.phase), left(T.left), nextp(T.nextp),
          tstate(T.tstate), cell_inv(T.cell_inv) {}
    // does the host's table put the arrays where this view expects them?  (rs_create checks it: a mismatch is a build error)
    static RS_HD bool matches(const Lds &T) {
        return T.node.off == o_node && T.aux.off == o_aux && T.vnx.off == o_vnx && T.sc.off == o_sc && T.alive.off == o_alive && T.alive0.off == o_alive0 &&
               T.ls_h.off == o_ls_h && T.ls_lc.off == o_ls_lc && T.ls_mh.off == o_ls_mh && T.grid.off == o_grid && T.lcap == kcap;
    }
};
RS_CARVE bool lds_fix_matches(const Lds &T, int C) {
    switch (C) {
        case 128: return LdsFix<128>::matches(T);
        case 256: return LdsFix<256>::matches(T);
        case 512: return LdsFix<512>::matches(T);
        case 768: return LdsFix<768>::matches(T);
        case 896: return LdsFix<896>::matches(T);
        case 1024: return LdsFix<1024>::matches(T);
        default: return true;           // (the kernel of a run-time capacity reads every offset from the table)
    }
}

// The grid cell length of a scenario: the shortest of CELL_CHOICES that can count the scenario's vehicles (PackedTables::cell_len_ok)
// and with which the working memory of an environment lets four workgroups share a CU; failing that, three; failing that the
// shortest countable one (resco_tables.h).  n_arr / n_dep / tls_maxl as PackedTables::build found them (they do not depend on it)
RS_CARVE float pick_cell_len(const rs_scenario *sc, int n_arr, int n_dep, int tls_maxl) {
    const float choices[] = CELL_CHOICES;
    const int n = (int)(sizeof(choices) / sizeof(choices[0]));
    const size_t limits[2] = {RS_LDS_4WG_LIMIT, RS_LDS_3WG_LIMIT};
    for (int k = 0; k < 2; ++k)
        for (int i = 0; i < n; ++i)
            if (PackedTables::cell_len_ok(sc, choices[i]) &&
                lds_carve(nullptr, sc->capacity, PackedTables::count_cells(sc, choices[i]), n_arr, n_dep, sc->n_obs, sc->n_signals, sc->n_vtypes, tls_maxl) <= limits[k])
                return choices[i];
    for (int i = 0; i < n; ++i) if (PackedTables::cell_len_ok(sc, choices[i])) return choices[i];
    return choices[0];          // (PackedTables::build reports the vehicle type that is too short)
}

// ------------------------------------------------------------------------------------------------ grid primitives
#ifdef RS_EMU_DEBUG         // host emulation, debug build: a chain that does not end means a vehicle entered a grid twice
#define RS_CHAIN_GUARD if (++rs_dbg_chain > 50000000L) { printf("endless chain walk\n"); fflush(stdout); abort(); }
#else
#define RS_CHAIN_GUARD
#endif
#ifndef RS_ASSERT
#define RS_ASSERT(c)
#endif
#ifndef RS_SEC             // section timers of a study build (resco_sim.hip)
#define RS_SEC_BEGIN
#define RS_SEC(id) {}
#endif
template <class LT> RS_DEV int lane_cells(const LT &L, const LaneRec &LR) { return (int)(LR.len * L.cell_inv) + 1; }
template <class LT> RS_DEV int cell_of(const LT &L, float pos, int ncell) { const int c = (int)(pos * L.cell_inv); return c < ncell ? c : ncell - 1; }

// THE grid as a tick sees it: the cells, and the tag (0 / CELL_TAG) of the contents this user means -- a reader the tag of its tick,
// the move phase that of the next one.  A cell that carries the other tag is empty for both.
struct Grid { uint16_t *c; uint32_t tag; };
RS_DEV unsigned long long grid_tag4(const Grid &g) { return g.tag ? 0x8000800080008000ull : 0ull; }
// head slot of the chain of cell c (NIL: the cell is empty or holds last tick's contents)
RS_DEV int cell_head(const Grid &g, int c) {
    const uint32_t v = g.c[c];
    return ((v ^ g.tag) & CELL_TAG) ? (int)NIL : (int)(v & NIL);
}
// push slot s into cell c; returns the previous head (the new chain link).  16-bit cells, exchanged with a CAS on the
// containing dword (LDS has no 16-bit atomics); the cell's vehicle count goes up by one
RS_DEV uint16_t grid_push(const Grid &g, int c, int s, bool mover) {
    uint32_t *w = (uint32_t *)g.c + (c >> 1);
    const int sh = (c & 1) * 16;
    const uint32_t flag = (mover ? CELL_MOVER : 0u) | g.tag;
    uint32_t old = *w, assumed, cell;
    do {
        assumed = old;
        cell = (assumed >> sh) & 0xFFFFu;
        if ((cell ^ g.tag) & CELL_TAG) cell = NIL;                // the other tag: last tick's contents -- an empty cell
        const uint32_t keep = cell & CELL_MOVER;                  // sticky mover flag of the cell
        uint32_t cnt = (cell >> CELL_CNT_SHIFT) & CELL_CNT_MAX;
        if (cnt < CELL_CNT_MAX) cnt += 1u;
        old = rs_atomic_cas(w, assumed, (assumed & ~(0xFFFFu << sh)) | (((uint32_t)s | (cnt << CELL_CNT_SHIFT) | keep | flag) << sh));
    } while (old != assumed);
    return (uint16_t)(cell & NIL);
}
// the 4 cells of the aligned quad starting at b, with the tag bit of a cell CLEAR iff it carries this user's tag
RS_DEV unsigned long long quad_load(const Grid &g, int b) { return *(const unsigned long long *)(g.c + b) ^ grid_tag4(g); }
// occupancy mask of the 4 cells of the aligned quad starting at b: bit 16 j + 11 set iff cell b + j is occupied
RS_DEV unsigned long long quad_occ(const Grid &g, int b) {
    const unsigned long long w = quad_load(g, b);
    const unsigned long long x = (w & 0x07FF07FF07FF07FFull) ^ 0x07FF07FF07FF07FFull;       // 11-bit field != 0: occupied
    return (x + 0x07FF07FF07FF07FFull) & 0x0800080008000800ull & ~(w >> 4);                 // ... and the tag is this tick's
}
// number of vehicles in the cells [c0, c0 + nc) (the lane with these cells): the sum of the cells' counters
RS_DEV int cells_count(const Grid &g, int c0, int nc) {
    const int c1 = c0 + nc - 1;
    int n = 0;
    for (int b = c0 & ~3; b <= c1; b += 4) {
        const unsigned long long w = quad_load(g, b);
        unsigned long long m = (w >> CELL_CNT_SHIFT) & 0x0007000700070007ull;
        m &= ~(((w >> 15) & 0x0001000100010001ull) * 7ull);     // cells of the other tag count nothing
        const int lo = c0 - b, hi = c1 - b;
        if (lo > 0) m &= ~0ull << (16 * lo);
        if (hi < 3) m &= ~0ull >> (16 * (3 - hi));
        n += (int)((m * 0x0001000100010001ull) >> 48);      // the four 3-bit counters added up in the top 16 bits
    }
    return n;
}
// first occupied cell of [c0, c1] scanning upwards, -1: none
RS_DEV int scan_up(const Grid &g, int c0, int c1) {
    for (int b = c0 & ~3; b <= c1; b += 4) {
        unsigned long long m = quad_occ(g, b);
        const int lo = c0 - b, hi = c1 - b;
        if (lo > 0) m &= ~0ull << (16 * lo);
        if (hi < 3) m &= ~0ull >> (16 * (3 - hi));
        if (m) return b + ((rs_ffsll(m) - 1) >> 4);
    }
    return -1;
}
// first occupied cell of [c0, c1] scanning downwards from c1, -1: none
RS_DEV int scan_down(const Grid &g, int c0, int c1) {
    for (int b = c1 & ~3; b + 3 >= c0; b -= 4) {
        unsigned long long m = quad_occ(g, b);
        const int lo = c0 - b, hi = c1 - b;
        if (lo > 0) m &= ~0ull << (16 * lo);
        if (hi < 3) m &= ~0ull >> (16 * (3 - hi));
        if (m) return b + ((63 - rs_clzll(m)) >> 4);
        if (b == 0) break;
    }
    return -1;
}
// any moving vehicle in the cells [c0, c0 + nc)?
RS_DEV bool cells_have_mover(const Grid &g, int c0, int nc) {
    const int c1 = c0 + nc - 1;
    for (int b = c0 & ~3; b <= c1; b += 4) {
        const unsigned long long w = quad_load(g, b);
        unsigned long long m = w & 0x4000400040004000ull & ~(w >> 1);
        const int lo = c0 - b, hi = c1 - b;
        if (lo > 0) m &= ~0ull << (16 * lo);
        if (hi < 3) m &= ~0ull >> (16 * (3 - hi));
        if (m) return true;
    }
    return false;
}
// ---- the order of the vehicles of a lane as ONE integer: larger = further ahead.  "j is ahead of i" (oracle: ahead_of) -- the front position first,
// the smaller trip id on a tie -- is `order_key(pj, kj) > order_key(pi, ki)`: positions are never negative (no -0.0 either: a position is a
// sum of non-negative terms, a lane length, or what is left of a position beyond a lane's end), so the bit patterns of two positions
// compare like the positions, and two vehicles never share a key (the trip ids differ).  One 64-bit compare and three selects per chain
// element instead of a ladder of branches (round 6: the scalar pipe, not the VALU, is the nearer wall -- DESIGN.md section 4).
RS_DEV unsigned long long order_key(float pos, int trip) { return ((unsigned long long)(uint32_t)rs_float_as_int(pos) << 16) | (uint32_t)(0xFFFF - trip); }
RS_DEV float key_pos(unsigned long long key) { return rs_int_as_float((int)(uint32_t)(key >> 16)); }
// a vehicle's Node with ONE 16-byte read (left to itself the compiler reads `nxt` first and the other fields it needs one by one,
// behind the branch that uses them: two or three dependent LDS round trips per chain element)
#ifndef RS_KEEP4
#define RS_KEEP4(a, b, c, d)
#endif
#ifndef RS_OPAQUE_S         // "the compiler knows nothing about this (wave-uniform) value from here on" (device build: an empty asm)
#define RS_OPAQUE_S(x)
#endif
template <class LT> RS_DEV Node node_load(const LT &L, int s) {
    struct alignas(16) Raw { uint32_t a, b, c, d; } r;
    __builtin_memcpy(&r, (const Node *)L.node + s, 16);
    RS_KEEP4(r.a, r.b, r.c, r.d)
    Node n;
    n.pos = rs_int_as_float((int)r.a); n.speed = rs_int_as_float((int)r.b); n.trip = (uint16_t)(r.c & 0xFFFFu); n.nxt = (uint16_t)(r.c >> 16);
    n.vt = (uint8_t)(r.d & 0xFFu); n.fl = (uint8_t)((r.d >> 8) & 0xFFu); n.sfq = (uint16_t)(r.d >> 16);
    return n;
}
// the winner of a search over chains: its slot (NIL: none), its order key (-> position) and the rest of its Node that a plan wants
// (speed, vType) -- carried along by the selects, so that nobody has to read the winner's Node again (one LDS round trip less on the
// critical path of every plan; where they are not used the compiler drops the selects)
struct Found {
    int slot;
    unsigned long long key;
    uint32_t speed_bits, vtw;
    RS_MEM float pos() const { return key_pos(key); }
    RS_MEM float speed() const { return rs_int_as_float((int)speed_bits); }
    RS_MEM int vt() const { return (int)(vtw & 0xFFu); }
};
// one step of a chain walk: the Node of slot s (returned: the next slot of the chain); `take` decides from its key whether it replaces f
#define RS_CHAIN_STEP(f, s, COND)                                                                          \
    {                                                                                                      \
        RS_CHAIN_GUARD                                                                                     \
        const Node nd_ = node_load(L, s);                                                                  \
        const unsigned long long key = order_key(nd_.pos, nd_.trip);                                       \
        const bool take_ = (COND);                                                                         \
        (f).slot = take_ ? (s) : (f).slot; (f).key = take_ ? key : (f).key;                                \
        (f).speed_bits = take_ ? (uint32_t)rs_float_as_int(nd_.speed) : (f).speed_bits;                    \
        (f).vtw = take_ ? (uint32_t)nd_.vt : (f).vtw;                                                      \
        s = nd_.nxt;                                                                                       \
    }
// rear-most vehicle of a cell chain (min pos, ties -> larger trip)
template <class LT> RS_DEV Found chain_rearmost(const LT &L, int head) {
    Found f{(int)NIL, ~0ull, 0u, 0u};
    for (int s = head & NIL; s != NIL;) RS_CHAIN_STEP(f, s, key < f.key)
    return f;
}
// front-most vehicle of a cell chain (max pos, ties -> smaller trip)
template <class LT> RS_DEV Found chain_frontmost(const LT &L, int head) {
    Found f{(int)NIL, 0ull, 0u, 0u};
    for (int s = head & NIL; s != NIL;) RS_CHAIN_STEP(f, s, key > f.key)
    return f;
}
// rear-most vehicle of the lane with cells [cell0, cell0 + ncell) whose front is within `win` metres of the lane start
template <class LT> RS_DEV Found rearmost_within(const LT &L, const Grid &grid, int cell0, int ncell, float win) {
    Found f{(int)NIL, ~0ull, 0u, 0u};
    if (win < 0.0f) return f;
    const int c = scan_up(grid, cell0, cell0 + cell_of(L, win, ncell));
    if (c < 0) return f;
    f = chain_rearmost(L, cell_head(grid, c));
    if (f.slot != NIL && f.pos() > win) f.slot = NIL;
    return f;
}
// nearest vehicle ahead of (pos, k) on the lane, at most `win` metres away (front to front).  (pos, k) are `self`'s own everywhere this
// is called (the plan; the lane-change searches on the neighbour lane, where `self` is not in the chains at all): its key equals
// `mykey`, so the strict comparison leaves it out without a test of its own.
template <class LT> RS_DEV Found leader_found(const LT &L, const Grid &grid, int cell0, int ncell, float pos, int k, int self, float win) {
    const int c = cell_of(L, pos, ncell);
    const unsigned long long mykey = order_key(pos, k);
    Found f{(int)NIL, ~0ull, 0u, 0u};
    // my own cell first.  A cell that holds ONE vehicle, me, needs no walk (the cell carries the number of its vehicles)
    const uint32_t cw = grid.c[cell0 + c];
    int s = ((cw ^ grid.tag) & CELL_TAG) ? (int)NIL : (int)(cw & NIL);
    if (s == self && ((cw >> CELL_CNT_SHIFT) & CELL_CNT_MAX) == 1u) s = NIL;
    while (s != NIL) {
        RS_ASSERT(s != self || order_key(L.node[s].pos, L.node[s].trip) == mykey)
        RS_CHAIN_STEP(f, s, (key > mykey) & (key < f.key))
    }
    if (f.slot == NIL && c + 1 < ncell) {
        const int cc = scan_up(grid, cell0 + c + 1, cell0 + cell_of(L, pos + win, ncell));
        if (cc >= 0) f = chain_rearmost(L, cell_head(grid, cc));
    }
    if (f.slot != NIL && f.pos() - pos > win) f.slot = NIL;
    return f;
}
template <class LT> RS_DEV int leader_within(const LT &L, const Grid &grid, int cell0, int ncell, float pos, int k, int self, float win) {
    return leader_found(L, grid, cell0, ncell, pos, k, self, win).slot;
}
// nearest vehicle behind (pos, k) on the lane (not `self`: see leader_found), at most `win` metres away
template <class LT> RS_DEV int follower_within(const LT &L, const Grid &grid, int cell0, int ncell, float pos, int k, int self, float win) {
    const int c = cell_of(L, pos, ncell);
    const unsigned long long mykey = order_key(pos, k);
    Found f{(int)NIL, 0ull, 0u, 0u};
    for (int s = cell_head(grid, cell0 + c); s != NIL;) {
        RS_ASSERT(s != self || order_key(L.node[s].pos, L.node[s].trip) == mykey)
        RS_CHAIN_STEP(f, s, (key < mykey) & (key > f.key))
    }
    if (f.slot == NIL && c > 0) {
        const float lo = pos - win;
        const int cc = scan_down(grid, cell0 + (lo > 0.0f ? cell_of(L, lo, ncell) : 0), cell0 + c - 1);
        if (cc >= 0) f = chain_frontmost(L, cell_head(grid, cc));
    }
    if (f.slot != NIL && pos - f.pos() > win) f.slot = NIL;
    (void)self;
    return f.slot;
}
// nearest vehicle of the lane whose front is at or behind `back`, at most `win` metres behind it
template <class LT> RS_DEV int at_or_behind_within(const LT &L, const Grid &grid, int cell0, int ncell, float back, float win) {
    if (back < 0.0f) return NIL;
    const int c = cell_of(L, back, ncell);
    Found f{(int)NIL, 0ull, 0u, 0u};
    for (int s = cell_head(grid, cell0 + c); s != NIL;) RS_CHAIN_STEP(f, s, !(nd_.pos > back) & (key > f.key))
    if (f.slot == NIL && c > 0) {
        const float lo = back - win;
        const int cc = scan_down(grid, cell0 + (lo > 0.0f ? cell_of(L, lo, ncell) : 0), cell0 + c - 1);
        if (cc >= 0) f = chain_frontmost(L, cell_head(grid, cc));
    }
    if (f.slot != NIL && back - f.pos() > win) f.slot = NIL;
    return f.slot;
}

// ------------------------------------------------------------------------------------------------ model helpers
template <class LT> RS_DEV int tls_state(const KTab &T, const LT &L, int tls, int pos) {
    if (tls == 0xFF) return TLS_G;
    return L.tstate[tls * T.tls_maxl + pos];
}
// the link a vehicle on lane `lane` (record LR) takes at route step rq: link index | NLINK_ARR, NLINK_NONE when there is
// none (last edge / a dead end for this route).  For a normal lane it is a table of (route step, lane index, trip parity)
RS_DEV uint16_t cache_link(const KTab &T, const LaneRec &LR, int lane, int rq, int trip) {
    if (LR.link_cnt == 0) return NLINK_NONE;
    if (LR.flags & LF_INTERNAL) { const int l = LR.link_start; return (uint16_t)(l | (T.links()[l].arr_idx >= 0 ? NLINK_ARR : 0)); }
    return T.next_link()[((size_t)rq * T.kmax + (lane - (int)LR.edge_lane0)) * 2 + (trip & 1)];
}
template <class LT> RS_DEV bool foe_blocked(const KTab &T, const LT &L, const Grid &grid, const LinkRec &K) {
    for (int i = K.foe_start; i < K.foe_start + K.foe_cnt; ++i) {
        const FoeRec F = T.foes()[i];
        if (F.tls != 0xFF && tls_state(T, L, F.tls, F.tls_pos) == TLS_R) continue;
        if (F.arr_idx >= 0 && L.arr[F.arr_idx] < RM_FOE_GAP_Q) return true;
        if (F.via1_cell0 != 0xFFFF && cells_have_mover(grid, F.via1_cell0, F.via1_nc)) return true;   // a moving vehicle on
        if (F.via2_cell0 != 0xFFFF && cells_have_mover(grid, F.via2_cell0, F.via2_nc)) return true;   // the foe's junction lanes
    }
    return false;
}
// copy the link states of signal s in phase ph into the working memory (called by the thread that owns the signal)
template <class LT> RS_DEV void tls_refresh(const KTab &T, const LT &L, const KParams &P, int s, int ph) {
    // rows of the state tables are tls_maxl bytes (a multiple of 4, zero padded) and 4-byte aligned: the row is copied with
    // independent 32-bit loads (up to 8 of them in flight) -- all signals change phase in the same ticks, and this copy is on the
    // critical path of those ticks' C phase
    const int w = T.tls_maxl >> 2;
    const uint32_t *src = (const uint32_t *)((P.fixed_program ? T.cold.fix8 + T.cold.fix_state_off[s] : T.cold.tls8 + T.cold.tls_state_off[s]) + ph * T.tls_maxl);
    uint32_t *dst = (uint32_t *)((uint8_t *)L.tstate + s * T.tls_maxl);
    uint32_t v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = src[i < w ? i : 0];
#pragma unroll
    for (int i = 0; i < 8; ++i) if (i < w) dst[i] = v[i];
#pragma unroll 1
    for (int i = 8; i < w; ++i) dst[i] = src[i];
}
template <class LT> RS_DEV void set_phase(const KTab &T, const LT &L, const KParams &P, int s, int ph) {
    if (ph < 0 || ph >= T.cold.tls_nphase[s]) return;
    L.phase[s] = ph;
    L.left[s] = P.tls_expiry ? T.cold.tls_dur[T.cold.tls_dur_off[s] + ph] : RM_TLS_HOLD_TICKS;      // rs_params.tls_expiry
    tls_refresh(T, L, P, s, ph);
}
// TLS switch events at the beginning of tick `tick` of this launch, preceded by Signal.set_phase when the yellow
// ticks are over
template <class LT> RS_DEV void tls_begin_of_tick(const KTab &T, const LT &L, const KParams &P, int s, int tick) {
    if (P.do_fsm && !P.fixed_program && tick == T.yellow_length) set_phase(T, L, P, s, L.nextp[s]);
    int left = L.left[s];
    if (left == 0) {
        const int32_t *dur = P.fixed_program ? T.cold.fix_dur + T.cold.fix_dur_off[s] : T.cold.tls_dur + T.cold.tls_dur_off[s];
        const int Pn = P.fixed_program ? T.cold.fix_nphase[s] : T.cold.tls_nphase[s];
        const int ph = (L.phase[s] + 1) % Pn;
        left = dur[ph];
        L.phase[s] = ph;
        tls_refresh(T, L, P, s, ph);
    }
    L.left[s] = left - 1;
}
// the continuation lengths of the lanes of route step rq (route_cont row), loaded once: the first four in registers
struct ContRow { float c[4]; const float *cn; };
RS_DEV ContRow cont_row(const KTab &T, int rq) {
    ContRow R;
    R.cn = T.route_cont() + (size_t)rq * T.kmax;
    for (int j = 0; j < 4; ++j) R.c[j] = j < T.kmax ? R.cn[j] : 0.0f;
    return R;
}
RS_DEV float cont_of(const ContRow &R, int j) {
    return j == 0 ? R.c[0] : (j == 1 ? R.c[1] : (j == 2 ? R.c[2] : (j == 3 ? R.c[3] : R.cn[j])));
}
// strategic lane-change need on lane index kk of an edge with n lanes (continuation lengths R), at position x with speed v:
// 0 when the lane is as good as any, or the need is still far away; else the direction of the nearest best lane.
// extra = RM_SG_EXTRA_LANES when asking whether a lane is good enough to move INTO for speed gain.  (oracle: strategic_dir_at)
// The vehicles on the lane to get to shorten the usable distance by occ_unit metres each [SUMO-K LC2013 best.occupation]:
// `grid` is the tick's grid and cell0 / nc the cell block of lane kk (the blocks of an edge's lanes are consecutive and
// equally sized).  OCC_NONE / OCC_FULL instead of a grid: assume that lane empty / full -- the bounds the scheduling hints
// are computed with while the next tick's grid is still being built.
#define OCC_NONE (Grid{(uint16_t *)0, 0u})
#define OCC_FULL (Grid{(uint16_t *)1, 0u})
// The table carries "lane k is NOT one of the best lanes of its route step" in the SIGN of the continuation length (PackedTables::build:
// the comparison `c >= best - RM_CONT_EPS` over the lanes of the step's edge is a property of the table, not of the vehicle): whoever
// wants the length takes the magnitude, and neither the maximum over the lanes nor the comparisons are evaluated per vehicle and tick.
RS_DEV bool cont_notbest(float c) { return rs_float_as_int(c) < 0; }
RS_DEV float cont_len(float c) { return rs_int_as_float(rs_float_as_int(c) & 0x7FFFFFFF); }
RS_DEV int strategic_dir(const ContRow &R, int kk, int n, float x, float v, int extra, float &rem, const Grid &grid, int cell0, int nc, float occ_unit) {
    const float mine = cont_of(R, kk);
    rem = cont_len(mine) - x;
#ifdef RS_EMU_CHECK_MAIL      // (host emulation: the sign is what the comparison over the lanes of THIS edge gives)
    { float best = 0.0f; for (int j = 0; j < n; ++j) { const float c = cont_len(cont_of(R, j)); if (c > best) best = c; }
      for (int j = 0; j < n; ++j) RS_ASSERT(cont_notbest(cont_of(R, j)) == !(cont_len(cont_of(R, j)) >= best - RM_CONT_EPS)) }
#endif
    if (!cont_notbest(mine)) return 0;
    int dl = 1000, dr = 1000;
    for (int j = kk + 1; j < n; ++j) if (!cont_notbest(cont_of(R, j))) { dl = j - kk; break; }
    for (int j = kk - 1; j >= 0; --j) if (!cont_notbest(cont_of(R, j))) { dr = kk - j; break; }
    const int dir = (dr <= dl) ? -1 : +1;
    if (grid.c == (uint16_t *)1) return dir;
    const int off = (dr <= dl ? dr : dl) + extra;
    const float la = (v > RM_LOOK_MIN_SPEED ? v : RM_LOOK_MIN_SPEED) * RM_LOOK_TIME + RM_LOOK_BASE;
    int cnt = 0;
    if (grid.c != (uint16_t *)0) cnt = cells_count(grid, cell0 + ((dr <= dl) ? -dr : dl) * nc, nc);
    if (rem - (float)cnt * occ_unit >= la * (float)off) return 0;
    return dir;
}
// approach registration for the coming tick: a moving vehicle whose next link somebody may have to yield to registers
// its arrival time there (v, pos, vType, lane length and next link of the vehicle AFTER this tick's move)
// (the three fields of the link the registration needs are one dword of its record: arr_idx | tls << 16 | tls_pos << 24)
RS_DEV uint32_t link_reg_word(const KTab &T, int nlk) { return ((const uint32_t *)&T.links()[nlk & 0x7FFF])[2]; }
template <class LT> RS_DEV void register_approach_w(const KTab &T, const LT &L, uint32_t kw, float v, float pos, float lane_len, int vt) {
    const int st = tls_state(T, L, (int)((kw >> 16) & 0xFFu), (int)(kw >> 24));
    if (st == TLS_R) return;
    const float dist = lane_len - pos;
    if (st == TLS_Y && dist >= d_brake_gap(v, L.vtp[vt * VT_COLS + VT_DECEL])) return;
    const float ta = RS_DIV(dist, v > 1.0f ? v : 1.0f);
    const int q = ta * 10.0f >= 65000.0f ? 65000 : (int)(ta * 10.0f);
    rs_atomic_min(&L.arr[(int16_t)(kw & 0xFFFFu)], q);
}
template <class LT> RS_DEV void register_approach(const KTab &T, const LT &L, int nlk, float v, float pos, float lane_len, int vt) {
    if (!(nlk & NLINK_ARR)) return;         // nobody yields to my next link (or I have none)
    if (v <= RM_HALT_SPEED) return;
    register_approach_w(T, L, link_reg_word(T, nlk), v, pos, lane_len, vt);
}
// follow `X` (a vehicle on a neighbouring lane of my edge) as if it were my leader, braking no harder than comfortably:
// the cooperative part of the lane changing (oracle: plan(), coop / coop_lead).  key = trip << 16 | slot.
template <class LT> RS_DEV void follow_neighbour(const KTab &T, const LT &L, uint32_t key, bool clamp_gap, const LaneRec &LR, int lane, float x, float v,
                             float b, float tau, float mingap, float &vsafe) {
    const int X = (int)(key & 0xFFFFu), kx = (int)(key >> 16);
    if ((int)L.node[X].trip != kx) return;
    const int lx = L.aux[X].lane;
    if (lx == (int)LANE_NONE || (LR.flags & LF_INTERNAL)) return;
    const int l0 = LR.edge_lane0, n = LR.flags >> 2;
    if (lx < l0 || lx >= l0 + n) return;            // not on my edge (lanes of an edge are consecutive; internal lanes have n = 0)
    const float *vo = L.vtp + L.node[X].vt * VT_COLS;
    const float px = L.node[X].pos, backx = px - vo[VT_LENGTH];
    if (clamp_gap ? !(px >= x) : !(backx >= x)) return;
    float g = backx - x - mingap;
    if (clamp_gap && g < 0.0f) g = 0.0f;
    float vs = d_follow_speed(g, L.node[X].speed, b, vo[VT_DECEL], tau);
    float vc = v - b; if (vc < 0.0f) vc = 0.0f;
    if (vs < vc) vs = vc;
    if (vs < vsafe) vsafe = vs;
    (void)lane;
}

// the mail flag of tick parity `par` in the record of slot s (see SFQ_MAIL): atomic, other threads may mark the same vehicle
template <class LT> RS_DEV void mail_flag_set(const LT &L, int s, int par) { rs_atomic_or((uint32_t *)((Node *)L.node + s) + 3, SFQ_MAIL(par) << 16); }
template <class LT> RS_DEV void mail_flag_clear(const LT &L, int s, int par) { rs_atomic_and((uint32_t *)((Node *)L.node + s) + 3, ~(SFQ_MAIL(par) << 16)); }
// mark slot s as one whose move of tick t is a long one and queue it (the plan's thread and the lane-change thread of a
// vehicle may both do it, at the same time: an atomic OR on the dword that holds Node.fl decides who queues it)
template <class LT, class LP> RS_DEV void list_push(const LT &L, const LP &list, int counter, int s);
template <class LT> RS_DEV void flag_mover(const LT &L, int s, int t, int lct = 0) {
    uint32_t *w = (uint32_t *)((Node *)L.node + s) + 3;
    const uint32_t bit = (uint32_t)fl_mh(t) << 8;
    if (!(rs_atomic_fetch_or(w, bit | ((uint32_t)lct << 8)) & bit)) list_push(L, L.ls_mh, SC_NMH + (t & 1), s);
}
// queue slot s in a work list (best effort: the flag in Node.fl is what counts, see the phases)
template <class LT, class LP> RS_DEV void list_push(const LT &L, const LP &list, int counter, int s) {
    const int i = rs_wave_ticket(&L.sc[counter]);
    if (i < (int)L.lcap) list[i] = (uint16_t)s;
}
// how far the plan of a vehicle looks ahead (speed v on a lane with limit vmax, speed factor sf): its free speed and the
// distance within which a leader or a stop line matters
RS_DEV float plan_vfree(const float *vt, float v, float lane_vmax, float sf) {
    float vfree = v + vt[VT_ACCEL];
    const float vl = lane_vmax * sf;
    if (vl < vfree) vfree = vl;
    if (vt[VT_MAXSPEED] < vfree) vfree = vt[VT_MAXSPEED];
    return vfree;
}
// (classify() and phase_plan() both evaluate plan_vfree / plan_look, at different inline sites; the short paths of the plan and the move
// rely on the two giving the SAME bits -- a vehicle without FL_H never has `seen < look` in its plan.  That holds because the library is
// built with -ffp-contract=off (resco_amd/build.py): no site-dependent FMA fusion.  The host emulation asserts it (RS_ASSERT), and
// every HIP == oracle test would see a missed stop line as a position difference.)
RS_DEV float plan_look(const float *vt, float vfree) { return d_brake_gap(vfree, vt[VT_DECEL]) + vfree * vt[VT_TAU] + vt[VT_MINGAP] + 1.0f; }
// FL_H for a vehicle at x on a lane of length len: the end of the lane is inside its look-ahead
RS_DEV bool looks_beyond(const float *vt, float v, float x, float lane_len, float lane_vmax, float sf) {
    return lane_len - x < plan_look(vt, plan_vfree(vt, v, lane_vmax, sf));
}
// can the lane-change decision of tick t have an effect for a vehicle in this state?  (a superset, computed before the
// grid of tick t is complete: it is on a lane that is not a best one -- whether the need has arisen depends on the queue
// beside it --, or it is its turn to look for speed gain on a neighbour lane that may be good enough)
RS_DEV bool may_change_lanes_row(const ContRow &R, const LaneRec &LR, int lane, int k, float x, float v, int t) {
    const int n = LR.flags >> 2;
    if ((LR.flags & LF_INTERNAL) || n < 2) return false;
    const int kk = lane - (int)LR.edge_lane0;
    float rem;
    if (strategic_dir(R, kk, n, x, v, 0, rem, OCC_FULL, 0, 0, 0.0f) != 0) return true;      // any lane that is not a best one
    if ((((uint32_t)t >> 1) + (uint32_t)k) & 3u) return false;
    const int tk = kk + ((t & 1) ? -1 : +1);
    return tk >= 0 && tk < n && strategic_dir(R, tk, n, x, v, RM_SG_EXTRA_LANES, rem, OCC_NONE, 0, 0, 0.0f) == 0;
}
// the same from the not-best MASK of the route step (KTab::notbest: bit k = lane k is not one of the step's best lanes -- the signs of the
// continuation row as one 16-bit word): the move classifies every vehicle every tick, and all it needs of the row is two of these bits.
// The lengths themselves are fetched only when the neighbour lane of a speed-gain candidate is not a best lane either (rare).
RS_DEV bool may_change_lanes(const KTab &T, int rq, uint32_t nb, const LaneRec &LR, int lane, int k, float x, float v, int t) {
    const int n = LR.flags >> 2;
    if ((LR.flags & LF_INTERNAL) || n < 2) return false;
    const int kk = lane - (int)LR.edge_lane0;
    bool r;
    if ((nb >> kk) & 1u) r = true;
    else if ((((uint32_t)t >> 1) + (uint32_t)k) & 3u) r = false;
    else {
        const int tk = kk + ((t & 1) ? -1 : +1);
        if (tk < 0 || tk >= n) r = false;
        else if (!((nb >> tk) & 1u)) r = true;
        else { float rem; r = strategic_dir(cont_row(T, rq), tk, n, x, v, RM_SG_EXTRA_LANES, rem, OCC_NONE, 0, 0, 0.0f) == 0; }
    }
#ifdef RS_EMU_CHECK_MAIL      // (host emulation)
    RS_ASSERT(r == may_change_lanes_row(cont_row(T, rq), LR, lane, k, x, v, t))
#endif
    return r;
}
// The work of tick t for the vehicle in slot s (state as of the beginning of that tick): its flags, and it is queued
template <class LT> RS_DEV int classify(const KTab &T, const LT &L, int s, const float *vt, float v, float x, const LaneRec &LR, int lane, int rq, uint32_t nb, int k, float sf, int t) {
    int fl = 0;
    if (looks_beyond(vt, v, x, LR.len, LR.vmax, sf)) { fl |= FL_H; list_push(L, L.ls_h, SC_NH, s); }
    if (may_change_lanes(T, rq, nb, LR, lane, k, x, v, t)) { fl |= FL_LC; list_push(L, L.ls_lc, SC_NLC, s); }
    return fl;
}

// ------------------------------------------------------------------------------------------------ the phases
// P: plan (Krauss car-following + links) for slot s.  LONG = false: the short path, for a vehicle WITHOUT FL_H -- nothing beyond
// the end of its lane is inside its look-ahead (classify() decided that on the very state this plan sees; the host emulation
// checks it) --, so the walk over the links is not compiled into what the waves on the short path run.
template <bool LONG, class LT> RS_DEV void phase_plan(const KTab &T, const LT &L, const Grid &grid, const State &G, size_t eo, const KParams &P, int genv, int t, int s) {
    RS_SEC_BEGIN
    const Aux ax = L.aux[s];
    const int lane = ax.lane;
    if (lane == (int)LANE_NONE) return;
    const Node me = L.node[s];
    const int k = me.trip;
    const float *vt = L.vtp + me.vt * VT_COLS;
    const float a = vt[VT_ACCEL], b = vt[VT_DECEL], tau = vt[VT_TAU], mingap = vt[VT_MINGAP];
    const float v = me.speed, x = me.pos;
    const float sf = sf_of(me.sfq);
    const LaneRec LR0 = T.lanes()[lane];
    LaneRec LR = LR0;
    float vsafe = RM_BIGF;
    // The look-ahead only FINDS what limits the vehicle (a leader, or a stop line = a standing leader of zero length):
    // the Krauss safe speed is evaluated once, by all lanes together, after the walk.
    float tgap = 0.0f, tvl = 0.0f, tbl = b;
    bool have = false;
    if (LONG) RS_SEC(7)
    const float vfree = plan_vfree(vt, v, LR.vmax, sf);
    const float look = plan_look(vt, vfree);
    const Found ld = leader_found(L, grid, LR.cell0, lane_cells(L, LR), x, k, s, look + T.maxlen);
    const int lead = ld.slot;
    if (LONG) RS_SEC(8)
    bool found = false;
    if (lead != NIL) {
        const float *vo = L.vtp + ld.vt() * VT_COLS;
        tgap = ld.pos() - vo[VT_LENGTH] - x - mingap;
        tvl = ld.speed(); tbl = vo[VT_DECEL];
        have = true; found = true;
    }
    if (me.sfq & SFQ_MAIL((t + 1) & 1)) {   // cooperation: requests of the last lane-change phase (the mailboxes are in HBM: read only when flagged)
        const int q = (t + 1) & 1;
        const uint32_t c1 = G.coop(q)[eo + s], c2 = G.cooplead(q)[eo + s];
        if (c1 != COOP_NONE) { G.coop(q)[eo + s] = COOP_NONE; follow_neighbour(T, L, c1, false, LR, lane, x, v, b, tau, mingap, vsafe); }
        if (c2 != COOP_NONE) { G.cooplead(q)[eo + s] = COOP_NONE; follow_neighbour(T, L, c2, true, LR, lane, x, v, b, tau, mingap, vsafe); }
        mail_flag_clear(L, s, q);
    }
#ifdef RS_EMU_CHECK_MAIL      // (host emulation: no request may wait in an unflagged vehicle's mailboxes)
    else { RS_ASSERT(G.coop((t + 1) & 1)[eo + s] == COOP_NONE && G.cooplead((t + 1) & 1)[eo + s] == COOP_NONE) }
#endif
    if (LONG) RS_SEC(9)
    float seen = LR.len - x;
    RS_ASSERT(LONG || found || !(seen < look))
    if (LONG && !found && seen < look) {
        int rq = ax.rq;
        int link = (int)(ax.nlink & 0x7FFF);
        int cur_lane = lane;
        const float bgv = d_brake_gap(v, b);        // can I still stop in front of a red / yellow light?
        for (int hop = 0; hop < RM_MAX_HOPS; ++hop) {
            const bool cur_int = (LR.flags & LF_INTERNAL) != 0;
            if (hop > 0) link = (int)(cache_link(T, LR, cur_lane, rq, k) & 0x7FFF);
            bool stop_here = false;
            LinkRec K;
            if (link == NLINK_NONE) {
                // last edge of the route: free run to its end; otherwise a dead end: wait for a lane change
                if (!cur_int && T.rsteps()[rq].next_edge == 0xFFFF) break;
                stop_here = true;
            } else {
                K = T.links()[link];
                const int st = tls_state(T, L, K.tls, K.tls_pos);
                if (K.tls != 0xFF && (st == TLS_R || st == TLS_Y) && seen >= bgv) stop_here = true;
                if (!stop_here && !(K.flags & KF_CONT) && ((K.flags & KF_MINOR) || (K.tls != 0xFF && st == TLS_g))) {
                    // a minor link is approached ready to stop until the foe lanes can be seen
                    if (seen > RM_VIS_DIST) stop_here = true;
                    else if (K.foe_cnt > 0 && foe_blocked(T, L, grid, K)) stop_here = true;
                }
            }
            if (stop_here) {
                const float g = seen - RM_STOP_OFFSET;
                tgap = g > 0.0f ? g : 0.0f; tvl = 0.0f; tbl = b;       // d_follow_speed(g, 0, b, b) == d_stop_speed(g, b)
                have = true;
                break;
            }
            LR = K.dest;
            cur_lane = K.to_lane;
            {   // slow down in time for a lower speed limit on the next lane
                const float vnl = LR.vmax * sf;
                if (vnl < vfree) {
                    const float vs = d_free_speed(seen, vnl, b);
                    if (vs < vsafe) vsafe = vs;
                }
            }
            const Found od = rearmost_within(L, grid, LR.cell0, lane_cells(L, LR), look - seen + T.maxlen);
            if (od.slot != NIL) {
                const float *vo = L.vtp + od.vt() * VT_COLS;
                tgap = seen + od.pos() - vo[VT_LENGTH] - mingap;
                tvl = od.speed(); tbl = vo[VT_DECEL];
                have = true;
                break;
            }
            if (!cur_int) rq += 1;
            seen += LR.len;
            if (!(seen < look)) break;
        }
    }
    if (LONG) RS_SEC(10)
    if (have) {
        const float vs = d_follow_speed(tgap, tvl, b, tbl, tau);
        if (vs < vsafe) vsafe = vs;
    }
    float vmin_n = v - b; if (vmin_n < 0.0f) vmin_n = 0.0f;
    float vmin_e = v - vt[VT_EMERGENCY]; if (vmin_e < 0.0f) vmin_e = 0.0f;
    const float lo = vsafe > vmin_e ? vsafe : vmin_e;
    const float vmin = vmin_n < lo ? vmin_n : lo;
    float vmax = vfree < vsafe ? vfree : vsafe;
    if (vmax < vmin) vmax = vmin;
    const float sigma = P.sigma >= 0.0f ? P.sigma : vt[VT_SIGMA];
    float vd = vmax;
    if (sigma > 0.0f) {
        const float r = d_u01(d_hash(P.seed, (uint32_t)genv, (uint32_t)k, (uint32_t)t, 0u));
        if (vd < a) vd -= sigma * vd * r; else vd -= sigma * a * r;
        if (vd < 0.0f) vd = 0.0f;
    }
    const float vnext = vd > vmin ? vd : vmin;
    L.vnx[s] = vnext;
#ifdef RS_EMU_DEBUG
    if (s == rs_dbg_slot && (t == rs_dbg_t || rs_dbg_t < 0))
        printf("plan t %d slot %d lane %d x %.3f v %.3f lead %d have %d tgap %.3f tvl %.3f vsafe %.3f vfree %.3f vnext %.3f nlink %x\n", t, s, lane, x, v, lead,
               (int)have, tgap, tvl, vsafe, vfree, vnext, ax.nlink);
#endif
    if (LONG) RS_SEC(3)
    if (x + vnext > LR0.len) flag_mover(L, s, t);
    if (LONG) RS_SEC(15)
    (void)LR;
}

// M: move slot s -- sideways first (the lane change decided in the plan phase), then forward, over to the next lanes, or
// out of the network; leave the old grid, enter the new one; register the approach of the coming tick
template <bool LONG, class LT> RS_DEV void phase_move(const KTab &T, const LT &L, const Grid &gnew, const State &G, const KParams &P, int env, size_t eo,
                       int t, bool last_tick, bool more, int s, int &active, int &halted, int &top) {
    const Aux ax = L.aux[s];
    const Node me = L.node[s];
    const int k = me.trip;
    int lane = ax.lane;
#ifdef RS_EMU_DEBUG
    {
        static int moved_at[65536];
        if (moved_at[s] == t + 1) { printf("slot %d moved twice in tick %d: fl %x lane %d trip %d\n", s, t, me.fl, lane, k); fflush(stdout); abort(); }
        moved_at[s] = t + 1;
    }
#endif
    LaneRec LR = T.lanes()[lane];
    // what the end of the move needs from the tables is requested now (the common case: the vehicle stays on its lane)
    uint32_t nb = T.notbest()[ax.rq];       // (which lanes of my route step are not its best ones: what classify() needs of the continuation row)
    uint32_t kw = 0;
    if (more && (ax.nlink & NLINK_ARR)) kw = link_reg_word(T, ax.nlink);
    const float sfv = sf_of(me.sfq);
    float tl = G.tloss()[eo + s];
    const int sw = ax.swait;
    int swn = sw;
    const float vn = L.vnx[s];
    int rq = ax.rq;
    uint16_t nlink = ax.nlink;              // the link after the lane the vehicle ends up on, as cache_link() returns it
    int link = (int)(nlink & 0x7FFF);
    bool relink = false;
    int side = 0;
    // LONG = false: a vehicle WITHOUT this tick's FL_MH -- its plan found that it stays on its lane and no lane change was decided --:
    // the hand-over, the arrival and the sideways move are not compiled into the code the waves on the short path run
    RS_ASSERT(LONG || (!(me.fl & (LCT_LEFT | LCT_RIGHT | LCT_SWAP_LEFT | LCT_SWAP_RIGHT)) && !(me.pos + vn > LR.len)))
    if (LONG && (me.fl & (LCT_LEFT | LCT_RIGHT | LCT_SWAP_LEFT | LCT_SWAP_RIGHT))) {
        if ((me.fl & (LCT_LEFT | LCT_RIGHT)) && !(me.pos + vn > LR.len)) side = (me.fl & LCT_LEFT) ? +1 : -1;
        else if (me.fl & (LCT_SWAP_LEFT | LCT_SWAP_RIGHT)) side = (me.fl & LCT_SWAP_LEFT) ? +1 : -1;
    }
    if (LONG && side) {
        lane += side;
        LR = T.lanes()[lane];
        nlink = cache_link(T, LR, lane, rq, k);
        link = (int)(nlink & 0x7FFF);
        relink = true;
    }
    const float vref = LR.vmax * sfv;
    if (last_tick && (P.out_mask & OUT_VEH_ACCEL)) G.accel()[eo + s] = vn - me.speed;
    if (vn <= RM_HALT_SPEED) {
        if (sw < 65535) swn = sw + 1;
        halted += 1;
        if (G.trip_log) { const int wt = G.wtot()[eo + s]; if (wt < 65535) G.wtot()[eo + s] = (uint16_t)(wt + 1); }
    } else swn = 0;
    if (vref > 0.0f && vn < vref) { tl += RS_DIV(vref - vn, vref); G.tloss()[eo + s] = tl; }
    float x = me.pos + vn;
    bool arrived = false;
    if (LONG) for (int it = 0; it < 16; ++it) {
        if (!(x > LR.len)) break;
        const bool li = (LR.flags & LF_INTERNAL) != 0;
        if (link == NLINK_NONE) {
            if (!li && T.rsteps()[rq].next_edge == 0xFFFF) arrived = true; else x = LR.len;
            break;
        }
        x -= LR.len;
        if (!li) rq += 1;
        {
            const LinkRec Km = T.links()[link];
            lane = Km.to_lane;
            LR = Km.dest;
        }
        nlink = cache_link(T, LR, lane, rq, k);
        link = (int)(nlink & 0x7FFF);
        relink = true;
    }
    if (LONG && arrived) {
        Aux na = ax; na.lane = LANE_NONE; na.swait = 0;
        L.aux[s] = na;
        L.node[s].trip = TRIP_NONE; L.node[s].fl = (uint8_t)(me.fl & fl_mh(t));
        // (the mailboxes of a free slot may keep a request nobody read: no request is addressed to a free slot, and the insertion
        //  empties all four -- four scattered stores per arrival less)
        rs_atomic_and(&L.alive[s >> 5], ~(1u << (s & 31)));
        {   // Signal.departures of the signal that observed the vehicle last (traffic_signal.py:226-232)
            const int ow = G.owner()[eo + s];
            if (ow != (int)OWNER_NONE) rs_atomic_add(&L.sig_dep[ow], 1);
        }
        G.owner()[eo + s] = OWNER_NONE; G.rwait()[eo + s] = 0;
        rs_atomic_add(&L.sc[SC_NACT], -1);
        rs_atomic_add(&L.sc[SC_STATS + ST_ARRIVED], 1);
        rs_atomic_add(&L.sc[SC_STATS + ST_DURATION], t + 1 - (int)G.depart()[eo + s]);
        rs_atomic_add(&L.sc[SC_STATS + ST_TLOSS], (int)(tl * 1024.0f + 0.5f));
        if (G.trip_log) {
            int32_t *r = G.trip_log + ((size_t)env * T.n_trips + k) * 4;
            r[0] = (int)G.depart()[eo + s]; r[1] = t + 1; r[2] = (int)(tl * 1024.0f + 0.5f); r[3] = (int)G.wtot()[eo + s];
        }
        return;
    }
    active += 1;
    if (s + 1 > top) top = s + 1;
    Aux na = ax;
    na.lane = (uint16_t)lane; na.rq = (uint16_t)rq; na.swait = (uint16_t)swn;
    if (LONG && relink) na.nlink = nlink;           // (looked up when the lane was entered: not loaded twice)
    L.aux[s] = na;
    Node nn = me;
    nn.pos = x; nn.speed = vn; nn.fl = (uint8_t)(me.fl & fl_mh(t));
    nn.nxt = grid_push(gnew, LR.cell0 + cell_of(L, x, lane_cells(L, LR)), s, vn > RM_HALT_SPEED);
    if (more) {
        if (LONG && relink) { nb = T.notbest()[rq]; if (na.nlink & NLINK_ARR) kw = link_reg_word(T, na.nlink); }
        nn.fl |= classify(T, L, s, L.vtp + me.vt * VT_COLS, vn, x, LR, lane, rq, nb, k, sfv, t + 1);
        L.node[s] = nn;
        if ((na.nlink & NLINK_ARR) && vn > RM_HALT_SPEED) register_approach_w(T, L, kw, vn, x, LR.len, me.vt);
    } else L.node[s] = nn;
    (void)P;
}

// the vehicle on the lane with cells [cell0, cell0 + ncell) whose body overlaps the one at (pos, k) lengthwise (the nearer one
// ahead first), NIL: none
template <class LT> RS_DEV int overlapping(const LT &L, const Grid &grid, int cell0, int ncell, float pos, int k, int self, float len_self) {
    const int lead = leader_within(L, grid, cell0, ncell, pos, k, self, RM_NB_WINDOW);
    if (lead != NIL && L.node[lead].pos - L.vtp[L.node[lead].vt * VT_COLS + VT_LENGTH] - pos < 0.0f) return lead;
    const int foll = follower_within(L, grid, cell0, ncell, pos, k, self, RM_NB_WINDOW);
    if (foll != NIL && pos - len_self - L.node[foll].pos < 0.0f) return foll;
    return NIL;
}

// The lane-change decision of slot s on the state at the beginning of the tick: returns LCT_* (0: stay).  A blocked strategic
// changer asks for cooperation; a mutual block is swapped out (oracle: lane_change())
template <class LT> RS_DEV int phase_lc_decide(const KTab &T, const LT &L, const Grid &grid, const State &G, size_t eo, int t, int s, const Aux &ax, const Node &me) {
    const int lane = ax.lane;
    // (everything the decision may need from global memory is requested at once)
    const LaneRec LR = T.lanes()[lane];
    const ContRow R = cont_row(T, ax.rq);
    const int my_wait = (int)ax.swait;
    const int n = LR.flags >> 2;
    if ((LR.flags & LF_INTERNAL) || n < 2) return 0;
    const int dir_allowed = (t & 1) ? -1 : +1;
    const int l0 = LR.edge_lane0, kk = lane - l0;
    const int k = me.trip;
    const float *vt = L.vtp + me.vt * VT_COLS;
    const float x = me.pos, v = me.speed;
    const int nc = lane_cells(L, LR);
    int want = 0, dir = dir_allowed;
    float rem;
    const int sdir = strategic_dir(R, kk, n, x, v, 0, rem, grid, LR.cell0, nc, T.occ_unit);
    if (sdir != 0) { dir = sdir; want = 2; }
    int code = 0;
    const int tk = kk + dir;
    if (tk >= 0 && tk < n) {
        const int tcell0 = (int)LR.cell0 + dir * nc;        // lanes of an edge own consecutive, equally sized cell blocks
        int lead_t = NIL, foll_t = NIL;
        bool have_t = false;
        if (!want && !((((uint32_t)t >> 1) + (uint32_t)k) & 3u)) {
            // speed gain between lanes that are both good: more room ahead on the neighbour.  A vehicle reconsiders only on
            // one pair of ticks (one left, one right chance) out of four
            float rem_t;
            if (strategic_dir(R, tk, n, x, v, RM_SG_EXTRA_LANES, rem_t, grid, tcell0, nc, T.occ_unit) == 0) {
                const int lead_c = leader_within(L, grid, LR.cell0, nc, x, k, s, RM_NB_WINDOW);
                if (lead_c != NIL) {
                    lead_t = leader_within(L, grid, tcell0, nc, x, k, s, RM_NB_WINDOW);
                    have_t = true;
                    const float gcur = L.node[lead_c].pos - L.vtp[L.node[lead_c].vt * VT_COLS + VT_LENGTH] - x;
                    float gtgt = RM_BIGF;
                    if (lead_t != NIL) gtgt = L.node[lead_t].pos - L.vtp[L.node[lead_t].vt * VT_COLS + VT_LENGTH] - x;
                    if (gcur < v * 3.0f + 15.0f && gtgt > gcur + RM_SG_ADVANTAGE) want = 1;
                }
            }
        }
        if (want) {
            if (!have_t) lead_t = leader_within(L, grid, tcell0, nc, x, k, s, RM_NB_WINDOW);
            foll_t = follower_within(L, grid, tcell0, nc, x, k, s, RM_NB_WINDOW);
            // urgent = strategic change close to the end of the drivable lane: accept tighter gaps (followers may have to
            // brake with their emergency deceleration), otherwise dense queues would never let anybody in
            const bool urgent = want == 2 && rem <= RM_URGENT_DIST;
            bool safe = true;
            if (lead_t != NIL) {
                const Node ld = L.node[lead_t];
                const float *vo = L.vtp + ld.vt * VT_COLS;
                const float gap = ld.pos - vo[VT_LENGTH] - x - (urgent ? 0.0f : vt[VT_MINGAP]);
                const float dec = urgent ? vt[VT_EMERGENCY] : vt[VT_DECEL];
                float vb = v - dec; if (vb < 0.0f) vb = 0.0f;
                if (gap < 0.0f || vb > d_follow_speed(gap, ld.speed, vt[VT_DECEL], vo[VT_DECEL], vt[VT_TAU])) safe = false;
            }
            if (safe && foll_t != NIL) {
                const Node fd = L.node[foll_t];
                const float *vo = L.vtp + fd.vt * VT_COLS;
                const float gap = x - vt[VT_LENGTH] - fd.pos - (urgent ? 0.0f : vo[VT_MINGAP]);
                const float dec = urgent ? vo[VT_EMERGENCY] : vo[VT_DECEL];
                float vb = fd.speed - dec; if (vb < 0.0f) vb = 0.0f;
                if (gap < 0.0f || vb > d_follow_speed(gap, v, vo[VT_DECEL], vt[VT_DECEL], vo[VT_TAU])) safe = false;
            }
            if (safe) {
                // (a vehicle that leaves its lane in this tick changes lanes on the next edge, if at all -- its plan looked at
                //  the links of the lane it is on: the move phase, which knows the next speed, drops the change then)
                if (dir == dir_allowed) code = dir > 0 ? LCT_LEFT : LCT_RIGHT;
            } else if (want == 2) {
                // blocked: fall in behind the target-lane leader, and ask the nearest vehicle completely behind me on the
                // target lane to let me in
                if (lead_t != NIL) { G.cooplead(t & 1)[eo + s] = ((uint32_t)L.node[lead_t].trip << 16) | (uint32_t)lead_t; mail_flag_set(L, s, t & 1); }
                const int R = at_or_behind_within(L, grid, tcell0, nc, x - vt[VT_LENGTH], RM_COOP_RANGE);
                if (R != NIL) { rs_atomic_min(&G.coop(t & 1)[eo + R], ((uint32_t)k << 16) | (uint32_t)s); mail_flag_set(L, R, t & 1); }
            }
        }
    }
    // Mutual block: two vehicles that have stood side by side near the end of their lanes for RM_SWAP_WAIT seconds, each in
    // the lane the other one needs, can never find a gap: they trade places.  The test is symmetric, so both threads reach
    // the same verdict (whichever way this tick's changes go).
    if (sdir != 0 && (t % RM_SWAP_EVERY) == 0 && v <= RM_HALT_SPEED && rem <= RM_URGENT_DIST &&
        my_wait >= RM_SWAP_WAIT) {
        const int b = overlapping(L, grid, (int)LR.cell0 + sdir * nc, nc, x, k, s, vt[VT_LENGTH]);
        if (b != NIL) {
            const Node nb = L.node[b];
            float rem_b;
            if (nb.speed <= RM_HALT_SPEED && (int)L.aux[b].swait >= RM_SWAP_WAIT &&
                strategic_dir(cont_row(T, L.aux[b].rq), kk + sdir, n, nb.pos, nb.speed, 0, rem_b, grid, (int)LR.cell0 + sdir * nc, nc, T.occ_unit) == -sdir && rem_b <= RM_URGENT_DIST &&
                overlapping(L, grid, LR.cell0, nc, nb.pos, nb.trip, b, L.vtp[nb.vt * VT_COLS + VT_LENGTH]) == s)
                code |= sdir > 0 ? LCT_SWAP_LEFT : LCT_SWAP_RIGHT;
        }
    }
        return code;
}

// C: does the oldest waiting trip of departure lane d get onto the network in tick t?  The space on its lane is judged
// AFTER this tick's move of the vehicles that are on it now -- their next speeds are known (oracle: insertion_check)
template <class LT> RS_DEV bool phase_insert_decide(const KTab &T, const LT &L, const Grid &grid, int t, int d) {
    if ((int)L.dep_t[d] > t) return false;           // nothing due on this lane (the common case: no global access)
    const int k = L.dep[d];
    // (one 8-byte record per departure lane instead of lane id -> lane record: the two loads of this check are independent)
    const DepInfo LR = T.cold.dep_info[d];
    const float *vt = L.vtp + T.trip_vtype()[k] * VT_COLS;
    const float mypos = vt[VT_LENGTH] < LR.len ? vt[VT_LENGTH] : LR.len;
    // only vehicles with pos - length < mypos + minGap can be in the way
    const int nc = (int)(LR.len * L.cell_inv) + 1;      // lane_cells(L, )
    const int c1 = LR.cell0 + cell_of(L, mypos + vt[VT_MINGAP] + T.maxlen, nc);
    for (int c = scan_up(grid, LR.cell0, c1); c >= 0; c = (c < c1 ? scan_up(grid, c + 1, c1) : -1))
        for (int o = cell_head(grid, c); o != NIL;) { RS_CHAIN_GUARD
            const Node od = L.node[o];
            const float back = (od.pos + L.vnx[o]) - L.vtp[od.vt * VT_COLS + VT_LENGTH];
            if (back - mypos - vt[VT_MINGAP] < 0.0f) return false;
            o = od.nxt;
        }
    return true;
}

// the r-th (0-based) free slot in ascending order, -1: none
template <class LT> RS_DEV int nth_free_slot(const LT &L, int C, int r) {      // (in the occupancy snapshot of the beginning of the tick)
#pragma unroll 1
    for (int w = 0; w < (C + 31) / 32; ++w) {
        uint32_t fr = ~L.alive0[w];
        if (w == (C - 1) / 32 && (C & 31)) fr &= (1u << (C & 31)) - 1u;
        const int c = rs_popc(fr);
        if (r >= c) { r -= c; continue; }
#pragma unroll 1
        for (int j = 0; j < r; ++j) fr &= fr - 1u;      // drop the r lowest set bits
        return w * 32 + rs_ffs(fr) - 1;
    }
    return -1;
}

// the lane-change decision of the vehicle in slot s, queued for the move when it changes lanes (what a lane-change chunk does per entry)
template <class LT> RS_DEV void lc_decide_and_flag(const KTab &T, const LT &L, const Grid &grid, const State &G, size_t eo, int t, int s) {
    const Aux ax = L.aux[s];
    if (ax.lane == LANE_NONE) return;
    const int code = phase_lc_decide(T, L, grid, G, eo, t, s, ax, L.node[s]);
    if (code) flag_mover(L, s, t, code);
}
// what phase_move adds to its thread's counters, as one word (a function that is CALLED returns it in a register): bit 0 the vehicle is
// still on the network, bit 1 it stands, bits 2.. its slot + 1 (0: it has arrived)
RS_DEV void move_unpack(uint32_t r, int &active, int &halted, int &top) {
    active += (int)(r & 1u); halted += (int)((r >> 1) & 1u);
    if ((int)(r >> 2) > top) top = (int)(r >> 2);
}
// The LONG code paths of a tick as the execution interface sees them (ex.plan_long / ex.lc_decide / ex.move_long): this default inlines
// them where they are used; a build with -DRS_CALL_LONG calls them as functions with a register allocation of their own (resco_sim.hip:
// measured in round 6, 1-2 % slower, not adopted).
struct ExecInline {
    template <class LT> RS_MEM static void plan_long(const KTab &T, const LT &L, const Grid &grid, const State &G, size_t eo, const KParams &P, int genv, int, int t, int s) {
        phase_plan<true>(T, L, grid, G, eo, P, genv, t, s);
    }
    template <class LT> RS_MEM static void lc_decide(const KTab &T, const LT &L, const Grid &grid, const State &G, size_t eo, int, int t, int s) {
        lc_decide_and_flag(T, L, grid, G, eo, t, s);
    }
    template <class LT> RS_MEM static void move_long(const KTab &T, const LT &L, const Grid &gnew, const State &G, const KParams &P, int env, size_t eo, int t, bool last_tick, bool more,
                                                     int s, int &active, int &halted, int &top) {
        phase_move<true>(T, L, gnew, G, P, env, eo, t, last_tick, more, s, active, halted, top);
    }
};

// ------------------------------------------------------------------------------------------------ the step
// CAP: the slot capacity as a compile-time constant (0: read it from the tables at run time)
template <int CAP, class Exec, class LT>
RS_DEV void rs_step_body_on(Exec &ex, const LT &L, const KTab &T, const State &G, const Out &O, const KParams &P,
                            const int32_t *actions, int env) {
    const int B = ex.B;
    const int C = CAP ? CAP : T.capacity, S = T.n_signals, NO = T.n_obs;
    const int genv = P.env_base + env;
    const size_t eo = (size_t)env * C;
    const int n_ticks = P.n_ticks;

    uint16_t *const grid0 = L.grid;

    // ---- L0: scalars, tables, TLS, backlog heads
    ex.phase(0, [&](int tid) {
        if (tid < SC_STATS + ST_N) L.sc[tid] = tid < 4 ? G.env[env * 4 + tid] : 0;
        for (int i = tid; i < T.n_vtypes * VT_COLS; i += B) L.vtp[i] = T.cold.vtype_params[i];
        for (int i = tid; i < (int)(L.gstride >> 1); i += B) ((uint32_t *)grid0)[i] = 0x07FF07FFu;      // every cell empty (tag 0)
        for (int i = tid; i < T.n_arr; i += B) L.arr[i] = ARR_NONE;
        for (int i = tid; i < (C + 31) / 32; i += B) { L.alive[i] = 0u; L.mailw[i] = 0u; }
        for (int i = tid; i < (T.n_dep + 31) / 32; i += B) L.insm[i] = 0u;
        for (int i = tid; i < T.n_dep; i += B) {
            const int k = G.dep_next[(size_t)env * T.n_dep + i];
            L.dep[i] = (uint16_t)k;
            L.dep_t[i] = k == (int)TRIP_NONE ? (uint16_t)0xFFFF : (uint16_t)T.cold.trip_depart[k];
        }
        for (int i = tid; i < S; i += B) {
            const int ph = G.tls[(env * S + i) * TLS_W + 0];
            L.phase[i] = ph;
            L.left[i] = G.tls[(env * S + i) * TLS_W + 1];
            L.nextp[i] = G.tls[(env * S + i) * TLS_W + 2];
            L.sig_arr[i] = 0; L.sig_dep[i] = G.tls[(env * S + i) * TLS_W + 3];     // (vehicles that left since the last observe)
            tls_refresh(T, L, P, i, ph);
        }
    });
    // ---- L1: the slab (once per env-step); Signal.prep_phase for every signal (traffic_signal.py:176-184), then the TLS
    //          events of tick 0
    ex.phase(1, [&](int tid) {
        const int hw0 = L.sc[SC_HW];
        for (int s = tid; s < C; s += B) {
            uint16_t ln = LANE_NONE, tr = TRIP_NONE;
            if (s < hw0) { ln = G.lane()[eo + s]; tr = G.trip()[eo + s]; }
            Aux ax; ax.lane = ln; ax.rq = 0; ax.nlink = NLINK_NONE; ax.swait = 0;
            if (ln == LANE_NONE) { L.aux[s] = ax; L.node[s].trip = TRIP_NONE; continue; }
            const float sp = G.speed()[eo + s], x = G.pos()[eo + s];
            const int rq = (int)T.routes()[T.trip_route()[tr]].start + (int)G.cursor()[eo + s];
            const LaneRec LR0 = T.lanes()[ln];
            ax.rq = (uint16_t)rq;
            ax.nlink = cache_link(T, LR0, ln, rq, tr);
            ax.swait = G.swait()[eo + s];
            L.aux[s] = ax;
            Node nn; nn.pos = x; nn.speed = sp; nn.trip = tr; nn.vt = T.trip_vtype()[tr];
            // the speed factor is a function of (seed, environment, trip): recomputed here (four hashes per vehicle and ENV-STEP) instead of
            // read back -- 4 B per slot and step less from HBM; RS_BUF_VEH_SF is written at the insertion, for whoever reads it
            nn.fl = 0; nn.sfq = (uint16_t)speed_factor_q(P, genv, tr, L.vtp + nn.vt * VT_COLS);
            // a request of the last tick before this launch waits in the mailboxes: the first plan reads those of parity (t + 1) & 1
            if ((G.mail[(size_t)env * ((C + 31) / 32) + (s >> 5)] >> (s & 31)) & 1u) nn.sfq |= (uint16_t)SFQ_MAIL((L.sc[SC_T] + 1) & 1);
            nn.nxt = grid_push(Grid{grid0, 0u}, LR0.cell0 + cell_of(L, x, lane_cells(L, LR0)), s, sp > RM_HALT_SPEED);
            if (n_ticks > 0) nn.fl = (uint8_t)classify(T, L, s, L.vtp + nn.vt * VT_COLS, sp, x, LR0, ln, rq, T.notbest()[rq], tr, sf_of(nn.sfq), L.sc[SC_T]);
            L.node[s] = nn;
            rs_atomic_or(&L.alive[s >> 5], 1u << (s & 31));
        }
        for (int s = B - 1 - tid; s < S; s += B) {
            if (P.do_fsm && !P.fixed_program) {
                const int a = actions[env * S + s], cur = L.phase[s], Gn = T.cold.tls_ngreen[s];
                if (a < 0 || a >= T.cold.tls_nphase[s]) L.nextp[s] = cur;
                else {
                    L.nextp[s] = a;
                    if (cur != a && cur < Gn && a < Gn) {
                        const int y = T.cold.tls_yellow[T.cold.tls_yel_off[s] + cur * Gn + a];
                        if (y >= 0) set_phase(T, L, P, s, y);
                    }
                }
            }
            if (n_ticks > 0) tls_begin_of_tick(T, L, P, s, 0);
        }
    });
    // ---- L2: approach registration of the first tick (later ticks register while they move, see M)
    if (n_ticks > 0) ex.phase(2, [&](int tid) {
        const int hw = L.sc[SC_HW];
        for (int s = tid; s < hw; s += B) {
            const Aux ax = L.aux[s];
            if (ax.lane == LANE_NONE) continue;
            const Node me = L.node[s];
            register_approach(T, L, ax.nlink, me.speed, me.pos, T.lanes()[ax.lane].len, me.vt);
        }
    });

    // A tick = three phases on the state at its beginning (oracle: orc_tick):
    //   P  every vehicle: next speed (car following, links, foes, cooperation) and lane-change decision
    //   C  role threads: which departure lanes insert (knowing the next speeds), the TLS events of the NEXT tick; the approach
    //      registers and the per-tick scalars are reset (nobody reads them between P and M)
    //   M  every vehicle: sideways, forward, hand-over, arrival; it leaves the grid of this tick's parity (which is empty
    //      afterwards) and enters the other one together with the inserted vehicles; the next tick's approach registration
    int cur = 0;
    for (int tick = 0; tick < n_ticks; ++tick) {
        const int t = L.sc[SC_T], hw = L.sc[cur ? SC_HWNEW : SC_HW];
        // the grid as this tick reads it (the cells tagged with its parity) and as its move phase builds it for the next one
        const Grid gold{grid0, cur ? CELL_TAG : 0u}, gnew{grid0, cur ? 0u : CELL_TAG};
        const bool more = tick + 1 < n_ticks;
        const int lcap = (int)L.lcap;
        const int nwv = B >> 6;
        // Who does what inside a phase is decided per WAVE (a wave runs every branch any of its lanes takes, and what it does
        // one after the other adds up on the critical path of the phase): the first waves take the lists of the vehicles with
        // the long code paths, the others share the slots on the short path.
        ex.phase(4, [&](int tid) {
            const int wv = ex.wave_of(tid), ln = tid & 63;          // (the wave index is wave-uniform: the role variables stay scalar)
            const int nh = L.sc[SC_NH], nlc = L.sc[SC_NLC];
            const unsigned long long r0 = ex.role_begin();
            int role = 0;
            if (nh > lcap || nlc > lcap) {
                // a list overflowed: every thread handles its slots in full (the flags, not the lists, carry the meaning), in two passes
                for (int pass = 0; pass < 2; ++pass)
                    for (int w = tid; w < hw; w += B) {
                        if (pass && !(L.node[w].fl & FL_LC)) continue;
                        if (pass) ex.lc_decide(T, L, gold, G, eo, env, t, w);
                        else ex.plan_long(T, L, gold, G, eo, P, genv, env, t, w);
                    }
            } else {
                // The work of the phase in chunks, longest code path first: the look-ahead list (RS_LIST_CHUNK entries per chunk), the
                // lane-change list, then the slots on the short path (64 per chunk).  A wave takes the next chunk when it is done
                // with its last one, so the phase ends when the work is done, not when the slowest of three fixed roles is.
                const int hch = (nh + RS_H_CHUNK - 1) / RS_H_CHUNK, lch = (nlc + RS_LIST_CHUNK - 1) / RS_LIST_CHUNK;
                const int total = hch + lch + ((hw + 63) >> 6);
                for (int it = 0;; ++it) {
                    const int c = ex.next_chunk(&L.sc[SC_CHUNK_P], wv, it, nwv);
                    if (c >= total) break;
                    if (c < hch) {                                  // plan of a vehicle that looks beyond its lane
                        role = 1;
                        const int w = c * RS_H_CHUNK + ln;
                        if (ln < RS_H_CHUNK && w < nh) ex.plan_long(T, L, gold, G, eo, P, genv, env, t, L.ls_h[w]);
                    } else if (c < hch + lch) {                     // lane-change decision
                        role = 2;
                        const int w = (c - hch) * RS_LIST_CHUNK + ln;
                        if (ln < RS_LIST_CHUNK && w < nlc) ex.lc_decide(T, L, gold, G, eo, env, t, L.ls_lc[w]);
                    } else {                                        // plan on the short path
                        const int w = (c - hch - lch) * 64 + ln;
                        if (w < hw && !(L.node[w].fl & FL_H)) phase_plan<false>(T, L, gold, G, eo, P, genv, t, w);
                    }
                }
            }
            ex.role_end(role == 1 ? 7 : (role == 2 ? 8 : 9), r0);
            for (int i = tid; i < (T.n_dep + 31) / 32; i += B) L.insm[i] = 0u;
        });
        ex.phase(5, [&](int tid) {
            for (int i = tid; i < T.n_arr; i += B) L.arr[i] = ARR_NONE;
            for (int i = tid; i < (C + 31) / 32; i += B) L.alive0[i] = L.alive[i];
            // the cells that still carry the tag the move phase is about to push with hold what was valid a tick ago (nobody read
            // them in this tick): they are emptied now, so that every cell of that tag the move meets is one it has filled itself.
            // (Readers of this phase are not disturbed: the cells of their tag are written back as they are.)
            // (the stride is made opaque per tick: seen through, the compiler keeps `8 * B` in a vector register across the whole tick loop --
            //  in the 64-VGPR build that register lived in scratch, written once and re-loaded in every iteration of this loop: 8 MB of HBM
            //  writes per launch of 2048 environments, profiles/r06_pmc_scratch_mailflags.txt)
            int Bs = B;
            RS_OPAQUE_S(Bs)
            for (int i = tid; i < (int)(L.gstride >> 2); i += Bs) {
                unsigned long long *q = (unsigned long long *)grid0 + i;
                const unsigned long long w = *q;
                const unsigned long long st = ((w ^ grid_tag4(gold)) >> 15) & 0x0001000100010001ull;    // 1: the cell carries the other tag
                if (st) { const unsigned long long m = st * 0xFFFFull; *q = (w & ~m) | ((0x07FF07FF07FF07FFull | grid_tag4(gold)) & m); }
            }
            for (int d = B - 1 - tid; d < T.n_dep; d += B)
                if (phase_insert_decide(T, L, gold, t, d)) rs_atomic_or(&L.insm[d >> 5], 1u << (d & 31));
            if (more)
                for (int sg = B - 1 - tid; sg < S; sg += B) tls_begin_of_tick(T, L, P, sg, tick + 1);
            if (tid == 0) {
                L.sc[SC_T] = t + 1; L.sc[SC_STATS + ST_TICKS] += 1;
                L.sc[cur ? SC_HW : SC_HWNEW] = 0;               // the high-water mark the move phase builds
                L.sc[SC_ROOM] = C - L.sc[SC_NACT];
                L.sc[SC_NH] = 0; L.sc[SC_NLC] = 0;              // the move phase queues the next tick's lists
                L.sc[SC_NMH + ((t + 1) & 1)] = 0;               // the next plan phase queues the next tick's movers
                L.sc[SC_CHUNK_P] = 0; L.sc[SC_CHUNK_M] = 0;     // (nobody takes tickets during C)
            }
        });
        ex.phase(6, [&](int tid) {
            int active = 0, halted = 0, top = 0;
            // chunks again, longest first: the vehicles that leave their lane (list), the insertions (one chunk, when there are
            // any), then the slots
            const int wv = ex.wave_of(tid), ln = tid & 63;
            const int nmh = L.sc[SC_NMH + (t & 1)];
            int ich = 0;
            for (int i = 0; i < (T.n_dep + 31) / 32; ++i) if (L.insm[i]) ich = 1;
            const bool all = nmh > lcap;
            const int mch = all ? 0 : (nmh + RS_LIST_CHUNK - 1) / RS_LIST_CHUNK;
            const int total = mch + ich + ((hw + 63) >> 6);
            const unsigned long long r0 = ex.role_begin();
            bool list = false;
            for (int it = 0;; ++it) {
                const int c = ex.next_chunk(&L.sc[SC_CHUNK_M], wv, it, nwv);
                if (c >= total) break;
                if (c < mch) {
                    list = true;
                    const int w = c * RS_LIST_CHUNK + ln;
                    if (ln < RS_LIST_CHUNK && w < nmh) ex.move_long(T, L, gnew, G, P, env, eo, t, !more, more, L.ls_mh[w], active, halted, top);
                } else if (c < mch + ich) {
                    // the winners of the departure lanes take the slots that were free at the beginning of the tick, lower lane first
                    for (int d = 63 - ln; d < T.n_dep; d += 64) {
                        if (!(L.insm[d >> 5] & (1u << (d & 31)))) continue;
                        int rank = rs_popc(L.insm[d >> 5] & ((1u << (d & 31)) - 1u));
                        for (int w = 0; w < (d >> 5); ++w) rank += rs_popc(L.insm[w]);
                        if (rank >= L.sc[SC_ROOM]) { rs_atomic_add(&L.sc[SC_STATS + ST_CAP_BLOCKED], 1); continue; }    // the network is full: counted, the trip stays in its backlog
                        const int s = nth_free_slot(L, C, rank);
                        if (s < 0) continue;
                        const int k = L.dep[d];
                        const int v = T.trip_vtype()[k];
                        const float *vt = L.vtp + v * VT_COLS;
                        const RouteRec RR = T.routes()[T.trip_route()[k]];
                        const LaneRec LRd = T.lanes()[RR.depart_lane];
                        const int sfq = speed_factor_q(P, genv, k, vt);
                        const float sfn = sf_of(sfq);
                        Node nn; nn.pos = vt[VT_LENGTH] < RR.depart_len ? vt[VT_LENGTH] : RR.depart_len;
                        nn.speed = 0.0f; nn.trip = (uint16_t)k; nn.vt = (uint8_t)v; nn.fl = 0; nn.sfq = (uint16_t)sfq;
                        nn.nxt = grid_push(gnew, LRd.cell0 + cell_of(L, nn.pos, lane_cells(L, LRd)), s, false);
                        if (more) nn.fl = (uint8_t)classify(T, L, s, vt, 0.0f, nn.pos, LRd, RR.depart_lane, (int)RR.start, T.notbest()[RR.start], k, sfn, t + 1);
                        L.node[s] = nn;
                        Aux na; na.lane = RR.depart_lane; na.rq = (uint16_t)RR.start; na.swait = 0;
                        na.nlink = cache_link(T, LRd, RR.depart_lane, (int)RR.start, k);
                        L.aux[s] = na;
                        G.sf()[eo + s] = sfn; G.tloss()[eo + s] = 0.0f; G.cooplead(0)[eo + s] = COOP_NONE; G.cooplead(1)[eo + s] = COOP_NONE; G.coop(0)[eo + s] = COOP_NONE; G.coop(1)[eo + s] = COOP_NONE;
                        G.rwait()[eo + s] = 0; G.owner()[eo + s] = OWNER_NONE; G.depart()[eo + s] = (uint16_t)(t + 1); G.accel()[eo + s] = 0.0f; G.wtot()[eo + s] = 0;
                        {
                            const int k2 = T.cold.trip_next[k];
                            L.dep[d] = (uint16_t)k2;
                            L.dep_t[d] = k2 == (int)TRIP_NONE ? (uint16_t)0xFFFF : (uint16_t)T.cold.trip_depart[k2];
                        }
                        rs_atomic_or(&L.alive[s >> 5], 1u << (s & 31));
                        rs_atomic_add(&L.sc[SC_NACT], 1);
                        rs_atomic_add(&L.sc[SC_NINS], 1);
                        rs_atomic_add(&L.sc[SC_STATS + ST_INSERTED], 1);
                        rs_atomic_add(&L.sc[SC_STATS + ST_DEPDELAY], t - T.cold.trip_depart[k]);
                        if (s + 1 > top) top = s + 1;
                        // (a standing vehicle does not register an approach)
                    }
                } else {
                    const int w = (c - mch - ich) * 64 + ln;
                    // (the vehicles of the list have FL_MH of this tick's parity set: their chunk moves them)
                    if (w < hw && (L.alive0[w >> 5] & (1u << (w & 31)))) {
                        if (all) ex.move_long(T, L, gnew, G, P, env, eo, t, !more, more, w, active, halted, top);
                        else if (!(L.node[w].fl & fl_mh(t))) phase_move<false>(T, L, gnew, G, P, env, eo, t, !more, more, w, active, halted, top);
                    }
                }
            }
            ex.role_end(list ? 10 : 3, r0);
            rs_wave_add(&L.sc[SC_STATS + ST_ACTIVE_TICKS], active);
            rs_wave_add(&L.sc[SC_STATS + ST_WAITING], halted);
            rs_wave_max(&L.sc[cur ? SC_HW : SC_HWNEW], top);
            ex.role_end(15, r0);
        });
        cur ^= 1;
    }
    const int hw_end = L.sc[cur ? SC_HWNEW : SC_HW];

    // ---- Signal.observe for every signal (traffic_signal.py:189-247); without it (step_sim, multi_signal.py:102-105) only the
    //      state goes back: the Signal objects' waiting times, owners and arrival / departure sets are left alone
    const bool obs = P.do_observe != 0;
    if (obs) ex.phase(11, [&](int tid) {
        for (int i = tid; i < NO; i += B) { L.agg_q[i] = 0; L.agg_a[i] = 0; L.agg_w[i] = 0; L.agg_m[i] = 0; L.agg_s[i] = 0; L.agg_n[i] = 0; }
    });
    ex.phase(12, [&](int tid) {
        const int hw0 = G.env[env * 4 + 2];
        const int top = hw_end > hw0 ? hw_end : hw0;
        int hi = 0;
        for (int s = tid; s < top; s += B) {
            const int lane = L.aux[s].lane;
            G.lane()[eo + s] = (uint16_t)lane; G.trip()[eo + s] = L.node[s].trip;
            if (lane == (int)LANE_NONE) continue;
            hi = s + 1;
            if (L.node[s].sfq & SFQ_MAIL((L.sc[SC_T] + 1) & 1)) rs_atomic_or(&L.mailw[s >> 5], 1u << (s & 31));      // (what the next launch's first plan reads)
            // store the slab back (once per env-step)
            const int rq = L.aux[s].rq;
            G.pos()[eo + s] = L.node[s].pos; G.speed()[eo + s] = L.node[s].speed; G.swait()[eo + s] = L.aux[s].swait;
            G.cursor()[eo + s] = (uint16_t)(rq - (int)T.routes()[T.trip_route()[L.node[s].trip]].start);
            if (!obs) continue;
            const int prev_owner = G.owner()[eo + s];
            const LaneRec LR = T.lanes()[lane];
            const int oi = T.cold.lane_obs[lane];
            bool detect = false;
            if (oi >= 0) {
                const float d = (LR.len - L.node[s].pos) + T.rsteps()[rq].tlsdist;
                detect = d <= P.max_distance;
            }
            if (!detect) {
                if (prev_owner != (int)OWNER_NONE) rs_atomic_add(&L.sig_dep[prev_owner], 1);
                G.owner()[eo + s] = OWNER_NONE; G.rwait()[eo + s] = 0;
                continue;
            }
            const int sig = T.cold.obs_sig[oi];
            int rw = G.rwait()[eo + s];
            if (prev_owner != sig) {
                rw = 0;
                rs_atomic_add(&L.sig_arr[sig], 1);
                rs_atomic_add(&L.agg_n[oi], 1);
                if (prev_owner != (int)OWNER_NONE) rs_atomic_add(&L.sig_dep[prev_owner], 1);
            }
            if (rw > 0) { rw += T.step_length; if (rw > 65535) rw = 65535; }
            else { const int sw = L.aux[s].swait; if (sw > 0) rw = sw; }
            G.rwait()[eo + s] = (uint16_t)rw;
            G.owner()[eo + s] = (uint8_t)sig;
            if (rw > 0) { rs_atomic_add(&L.agg_q[oi], 1); rs_atomic_add(&L.agg_w[oi], rw); rs_atomic_max(&L.agg_m[oi], rw); }
            else rs_atomic_add(&L.agg_a[oi], 1);
            rs_atomic_add((int32_t *)&L.agg_s[oi], (int32_t)(uint32_t)(L.node[s].speed * 65536.0f + 0.5f));
        }
        if (hi) rs_atomic_max(&L.sc[SC_HWOUT], hi);
    });
    // per observed lane rows, written as flat coalesced streams; states / rewards -- only the buffers the caller asked for
    // (rs_set_outputs); state write-back
    ex.phase(13, [&](int tid) {
        const uint32_t om = obs ? P.out_mask : 0u;
        if (om & (OUT_LANE_AGG | OUT_DRQ_NORM)) for (int i = tid; i < NO * 5; i += B) {
            const int oi = i / 5, c = i - oi * 5;
            const int sg = T.cold.obs_sig[oi];
            const float sp = (float)L.agg_s[oi] * (1.0f / 65536.0f);
            float raw, nrm;
            if (c == 0) { raw = (float)L.agg_q[oi]; nrm = (oi - T.cold.sig_obs_start[sg]) == L.phase[sg] ? 1.0f : 0.0f; }
            else if (c == 1) { raw = (float)L.agg_a[oi]; nrm = raw / 28.0f; }
            else if (c == 2) { raw = (float)L.agg_w[oi]; nrm = raw / 28.0f; }
            else if (c == 3) { raw = (float)L.agg_m[oi]; nrm = (float)L.agg_q[oi] / 28.0f; }
            else { raw = sp; nrm = sp / 20.0f / 28.0f; }
            if (om & OUT_LANE_AGG) O.lane_agg()[(size_t)env * NO * 5 + i] = raw;         // queue, approach, total_wait, max_wait, speed_sum
            if (om & OUT_DRQ_NORM) O.drq_norm()[(size_t)env * NO * 5 + i] = nrm;         // one-hot(lane position == phase), approach, wait, queue, speed
        }
        if (om & OUT_LANE_ARR) for (int i = tid; i < NO; i += B) O.lane_arr()[(size_t)env * NO + i] = L.agg_n[i];
        if (om & OUT_DRQ_F16) for (int i = tid; i < S * T.lmax * 5; i += B) {
            const int sg = i / (T.lmax * 5), r = i - sg * T.lmax * 5;
            const int l = r / 5, c = r - l * 5;
            const int o0 = T.cold.sig_obs_start[sg];
            float nrm = 0.0f;                                    // zero padding beyond the signal's lanes
            if (l < T.cold.sig_obs_start[sg + 1] - o0) {
                const int oi = o0 + l;
                if (c == 0) nrm = l == L.phase[sg] ? 1.0f : 0.0f;
                else if (c == 1) nrm = (float)L.agg_a[oi] / 28.0f;
                else if (c == 2) nrm = (float)L.agg_w[oi] / 28.0f;
                else if (c == 3) nrm = (float)L.agg_q[oi] / 28.0f;
                else nrm = (float)L.agg_s[oi] * (1.0f / 65536.0f) / 20.0f / 28.0f;
            }
            O.drq_f16()[(size_t)env * S * T.lmax * 5 + i] = rs_f2h(nrm);
        }
        // states.mplight / wave / mplight_full: one thread per (signal, movement)
        if (om & (OUT_MPLIGHT | OUT_WAVE | OUT_MPLIGHT_FULL)) for (int i = tid; i < S * 12; i += B) {
            const int sg = i / 12, m = i - sg * 12;
            int q = 0, wv = 0, tw = 0, ap = 0;
            float last_speed = 0.0f;
            for (int j = T.cold.mv_in_start[i]; j < T.cold.mv_in_start[i + 1]; ++j) {
                const int oi = T.cold.mv_in_idx[j];
                q += L.agg_q[oi]; wv += L.agg_q[oi] + L.agg_a[oi]; tw += L.agg_w[oi]; ap += L.agg_a[oi];
                last_speed = (float)L.agg_s[oi] * (1.0f / 65536.0f);       // states.py:97: total_speed restarts with every lane
            }
            for (int j = T.cold.mv_out_start[i]; j < T.cold.mv_out_start[i + 1]; ++j) q -= L.agg_q[T.cold.mv_out_idx[j]];
            const size_t so = (size_t)env * S + sg;
            if (om & OUT_MPLIGHT) O.mplight()[so * 13 + 1 + m] = q;
            if (om & OUT_WAVE) O.wave()[so * 12 + m] = wv;
            if (om & OUT_MPLIGHT_FULL) {
                float *mf = O.mplight_full() + so * 49 + 1 + m * 4;         // states.mplight_full (states.py:83-113)
                mf[0] = (float)q; mf[1] = (float)tw / 28.0f; mf[2] = last_speed; mf[3] = (float)ap / 28.0f;
            }
        }
        // per signal: phase, rewards, metrics
        for (int sg = tid; sg < S; sg += B) {
            const int ph = L.phase[sg];
            if (obs) {
                const int o0 = T.cold.sig_obs_start[sg], o1 = T.cold.sig_obs_start[sg + 1];
                int tw = 0, tq = 0, mq = 0;
                for (int oi = o0; oi < o1; ++oi) { tw += L.agg_w[oi]; const int qq = L.agg_q[oi]; tq += qq; if (qq > mq) mq = qq; }
                const size_t so = (size_t)env * S + sg;
                O.phase()[so] = ph; O.queue_sum()[so] = tq; O.queue_max()[so] = mq;
                O.wait()[so] = -(float)tw;
                const float wn = -(float)tw / 224.0f;
                O.wait_norm()[so] = wn < -4.0f ? -4.0f : (wn > 4.0f ? 4.0f : wn);
                int pr = tq;
                for (int i = T.cold.pr_out_start[sg]; i < T.cold.pr_out_start[sg + 1]; ++i) pr -= L.agg_q[T.cold.pr_out_idx[i]];
                O.pressure()[so] = -pr;
                if (om & OUT_MPLIGHT) O.mplight()[so * 13] = ph;
                if (om & OUT_MPLIGHT_FULL) O.mplight_full()[so * 49] = (float)ph;
                O.arrivals()[so] = L.sig_arr[sg]; O.departures()[so] = L.sig_dep[sg];
            }
            G.tls[(env * S + sg) * TLS_W + 0] = ph;
            G.tls[(env * S + sg) * TLS_W + 1] = L.left[sg];
            G.tls[(env * S + sg) * TLS_W + 2] = L.nextp[sg];
            G.tls[(env * S + sg) * TLS_W + 3] = obs ? 0 : L.sig_dep[sg];
        }
        for (int i = tid; i < T.n_dep; i += B) G.dep_next[(size_t)env * T.n_dep + i] = L.dep[i];
        for (int i = tid; i < (C + 31) / 32; i += B) G.mail[(size_t)env * ((C + 31) / 32) + i] = L.mailw[i];
    });
    ex.phase(14, [&](int tid) {
        if (tid == 0) {
            G.env[env * 4 + 0] = L.sc[SC_T]; G.env[env * 4 + 1] = L.sc[SC_NINS]; G.env[env * 4 + 3] = L.sc[SC_NACT];
            G.env[env * 4 + 2] = L.sc[SC_HWOUT];
        }
        if (tid < ST_N) {
            long long *st = G.stats + (size_t)env * ST_N;
            if (tid == ST_ACTIVE) st[tid] = L.sc[SC_NACT];
            else if (tid == ST_PENDING) {
                // trips whose insertion has been tried and failed so far: departed before the last tick, not on the network
                const int t = L.sc[SC_T];
                const int hz = t - 1 <= T.horizon ? t - 1 : T.horizon;
                st[tid] = (t >= 1 ? T.cold.trips_cum[hz] : 0) - L.sc[SC_NINS];
            } else st[tid] += L.sc[SC_STATS + tid];
        }
    });
}

// the step for a capacity known at compile time (CAP: the offsets that follow from it are literals) or at run time (CAP = 0)
template <int CAP, class Exec>
RS_DEV void rs_step_body(Exec &ex, const Lds &L, const KTab &T, const State &G, const Out &O, const KParams &P,
                         const int32_t *actions, int env) {
    if constexpr (CAP != 0) {
        const LdsFix<CAP> Lf(L);
        rs_step_body_on<CAP>(ex, Lf, T, G, O, P, actions, env);
    } else rs_step_body_on<CAP>(ex, L, T, G, O, P, actions, env);
}
