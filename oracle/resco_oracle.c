/*
 * resco_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).  See resco_oracle.h.
 *
 * What it restates and what pins it:
 *   - everything ABOVE the simulator (phase lists, create_yellows, prep / set FSM, Signal.observe, the RESCO waiting rule,
 *     states, rewards, arrivals / departures) follows resco_benchmark/traffic_signal.py and multi_signal.py line by line
 *     (citations at the functions) and is PINNED: the reference's unmodified Python runs over this oracle in the build
 *     container (oracle/ref_harness.py) and its outputs are the golden fixtures (tests/golden/, tests/test_oracle_golden.py);
 *   - simulationStep() itself is SUMO, a third-party binary that is neither vendored nor installable here: the
 *     microsimulation below is this build's OWN model after SUMO's published algorithms ([SUMO-K] in SURVEY.md).
 *     PARITY UNPINNED against SUMO; tools/sumo_runner.py holds the comparison for a box that has SUMO.
 *
 * Synchronous tick: every decision of a tick is taken on the state at its beginning and no phase reads what the same
 * phase writes, so a data-parallel implementation (one GPU lane per vehicle) must reproduce it bit-for-bit in any order.
 * All kinematics are IEEE fp32 with no fused contraction (compile with -ffp-contract=off).
 *
 * Per tick (orc_tick):
 *   TLS switch events -> lane lists -> link approach registration -> plan (Krauss, links, foes, cooperation) ->
 *   lane-change decision -> insertion check (with the planned speeds) -> move (sideways, forward, hand-over, arrival) ->
 *   insertion
 */
#include "resco_oracle.h"
#include "../include/resco_model.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define LANE_NONE 0xFFFFu
#define LANE_PENDING 0xFFFEu      /* (no longer a state: waiting trips hold no slot) kept so that `lane >= LANE_PENDING` reads "no vehicle" */
#define OWNER_NONE 0xFFu
#define NIL (-1)
#define ARR_NONE 65535
#define HALT_SPEED RM_HALT_SPEED
#define STOP_OFFSET RM_STOP_OFFSET
#define MAX_HOPS RM_MAX_HOPS
#define BIGF RM_BIGF

enum { VT_LENGTH, VT_MINGAP, VT_ACCEL, VT_DECEL, VT_TAU, VT_SIGMA, VT_MAXSPEED, VT_SF_MEAN, VT_SF_DEV, VT_EMERGENCY, VT_COLS };
enum { TLS_R = 0, TLS_Y = 1, TLS_g = 2, TLS_G = 3 };

struct orc_env {
    const orc_scenario *sc;
    orc_params p;
    int32_t env_index;
    int32_t t;              /* ticks since begin */
    float maxlen;           /* longest vehicle of the scenario */
    float occ_unit;         /* length + minGap of the scenario's most common vehicle type: what one queued vehicle occupies */
    int32_t *lane_cnt;      /* per lane: vehicles on it at the beginning of the tick (build_lists) */
    int32_t room_ins;       /* free capacity when this tick's insertions were decided */
    uint8_t *free_before;   /* per slot: free when this tick's insertions were decided */
    int32_t n_inserted;     /* trips inserted so far */
    int32_t n_active;       /* vehicles on the network */
    int32_t *dep_next;      /* per lane: the next trip that departs from it (-1: none left) -- a FIFO per departure lane */
    int32_t *trip_next;     /* trip -> next trip with the same departure lane */
    int32_t *dep_first;     /* per lane: its first trip */
    int32_t *coop_lead;     /* per slot: the vehicle on my strategic target lane I try to fall in behind (slot, -1 none), with its trip in coop_lead_trip */
    int32_t *coop_lead_trip;
    int64_t *coop;          /* per slot: cooperation request of this tick's lane-change phase, (trip << 32 | slot) of the changer, -1 none */
    int32_t hw;             /* high-water mark: slots >= hw are free */
    int32_t *trip;          /* slot -> trip index, -1 free */
    /* vehicles */
    uint16_t *lane, *cursor, *sumo_wait, *resco_wait, *depart;
    uint8_t *owner;
    float *pos, *speed, *accel, *time_loss, *vnext;
    int32_t *lc_target;
    uint16_t *wtot;
    int32_t *trip_log;
    int32_t *dbg_reason, *dbg_block;   /* why the last plan() limited each vehicle (debug aid) */
    /* lane lists */
    int32_t *lane_head, *next_in_lane;
    int32_t *link_arr;
    int32_t *lane_ins;
    /* signals */
    int32_t *phase, *left, *next_phase;
    /* outputs */
    float *lane_agg, *drq_norm, *wait, *wait_norm;
    int32_t *agg_q, *agg_a, *agg_w, *agg_m;
    uint32_t *agg_s;
    int32_t *out_phase, *mplight, *wave, *pressure, *queue_sum, *queue_max;
    int32_t *sig_arr, *sig_dep, *out_arr, *out_dep;     /* |Signal.arrivals| / |Signal.departures| of the running / last observe */
    int32_t *lane_arr;      /* [n_obs] vehicles of the lane that are in their signal's `arrivals` set (rewards.fma2c fringe arrivals) */
    float *mplight_full;
    int64_t stats[11];
};

/* ------------------------------------------------------------------ counter-based RNG (murmur3_32) */
static inline uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
uint32_t orc_hash(uint32_t seed, uint32_t env, uint32_t trip, uint32_t tick, uint32_t stream) {
    uint32_t h = seed;
    uint32_t w[4] = {env, trip, tick, stream};
    for (int i = 0; i < 4; ++i) {
        uint32_t k = w[i];
        k *= 0xcc9e2d51u; k = rotl32(k, 15); k *= 0x1b873593u;
        h ^= k; h = rotl32(h, 13); h = h * 5u + 0xe6546b64u;
    }
    h ^= 16u;
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}
static inline float u01(uint32_t h) { return (float)(h >> 8) * (1.0f / 16777216.0f); }

/* ------------------------------------------------------------------ Krauss arithmetic [SUMO-K]
 * MSCFModel::brakeGapEuler / maximumSafeStopSpeedEuler / maximumSafeFollowSpeed, dt = 1 s. */
float orc_brake_gap(float v, float b) {
    int steps = (int)(v / b);
    float fs = (float)steps;
    return fs * v - b * fs * (fs + 1.0f) * 0.5f;
}
float orc_stop_speed(float gap, float b, float tau) {
    float g = gap - 0.001f;
    if (g < 0.0f) return 0.0f;
    float q = 1.0f + 4.0f * ((2.0f * g / b - tau) + tau * tau);
    float n = floorf(0.5f - (tau + sqrtf(q) * -0.5f));
    float h = 0.5f * n * (n - 1.0f) * b + n * b * tau;
    float r = (g - h) / (n + tau);
    return n * b + r;
}
/* MSCFModel::freeSpeed (Euler): highest speed now that still allows reaching `target` after `dist` metres
 * when decelerating with b per step [SUMO-K] */
float orc_free_speed(float dist, float target, float b) {
    if (dist < target) return target;
    float t2 = b + 2.0f * target;
    float y = ((sqrtf(t2 * t2 + 8.0f * b * dist) - b) * 0.5f - target) / b;
    if (y < 0.0f) y = 0.0f;
    float yf = floorf(y);
    float exact = (yf * yf + yf) * 0.5f * b + yf * target + (y > yf ? target : 0.0f);
    float rest = dist - exact;
    if (rest < 0.0f) rest = 0.0f;
    return rest / (yf + 1.0f) + yf * b + target;
}
float orc_follow_speed(float gap, float vl, float b, float bl, float tau) {
    float bm = b > bl ? b : bl;
    return orc_stop_speed(gap + orc_brake_gap(vl, bm), b, tau);
}

/* ------------------------------------------------------------------ helpers */
static inline const float *vt_of(const orc_env *e, int32_t trip) {
    return e->sc->vtype_params + (size_t)e->sc->trip_vtype[trip] * VT_COLS;
}
static inline int32_t trip_of_slot(const orc_env *e, int32_t slot) { return e->trip[slot]; }
static float speed_factor(const orc_env *e, int32_t trip) {
    const float *vt = vt_of(e, trip);
    float f = vt[VT_SF_MEAN];
    if (e->p.speed_dev) {
        /* Irwin-Hall(4) normal surrogate: basic arithmetic only, identical on CPU and GPU */
        float s = 0.0f;
        for (uint32_t i = 0; i < 4; ++i) s += u01(orc_hash(e->p.seed, (uint32_t)e->env_index, (uint32_t)trip, 0xFFFFFFFFu, i));
        float z = (s - 2.0f) * 1.7320508f;
        f = vt[VT_SF_MEAN] + vt[VT_SF_DEV] * z;
    }
    if (f < 0.2f) f = 0.2f;
    if (f > 2.0f) f = 2.0f;
    return (float)(int32_t)(f * RM_SF_QUANT + 0.5f) * (1.0f / RM_SF_QUANT);       /* a multiple of 1 / 4096 */
}
static inline int ahead_of(float pj, int32_t kj, float pi, int32_t ki) { /* j strictly ahead of i */
    return pj > pi || (pj == pi && kj < ki);
}
/* the link a vehicle on `lane` takes at route position `cursor`: of the connections to the next route edge the one whose
 * destination lane lets it drive on furthest without a lane change (route_cont, SUMO's bestLanes [SUMO-K]); -1 when the
 * lane has no connection to the next edge, or only into an edge too short to change lanes on (a dead end for this
 * route: the vehicle waits here for a lane change) */
static int32_t choose_link(const orc_env *e, int32_t lane, int32_t route, int32_t cursor, int32_t trip) {
    const orc_scenario *sc = e->sc;
    int32_t ls = sc->lane_link_start[lane], lc = sc->lane_link_cnt[lane];
    if (lc == 0) return -1;
    if (sc->lane_internal[lane]) return ls;
    int32_t rs = sc->route_start[route], rn = sc->route_start[route + 1] - rs;
    if (cursor + 1 >= rn) return -1;
    int32_t ne = sc->route_edge[rs + cursor + 1];
    const float *cn = sc->route_cont + (size_t)(rs + cursor + 1) * sc->kmax;
    float bc = -1.0f;
    for (int32_t l = ls; l < ls + lc; ++l) {
        if (sc->link_to_edge[l] != ne) continue;
        float c = cn[sc->link_dest_lane[l] - sc->edge_lane0[ne]];
        if (c > bc) bc = c;
    }
    if (bc < RM_MIN_LC_LEN) return -1;
    /* parallel connections whose destination lanes are all good enough (the best one, or GOOD_CONT metres without a
     * lane change) share the traffic: trips alternate between them (SUMO spreads them by lane occupation [SUMO-K]) */
    int32_t n_acc = 0;
    for (int32_t l = ls; l < ls + lc; ++l) {
        if (sc->link_to_edge[l] != ne) continue;
        float c = cn[sc->link_dest_lane[l] - sc->edge_lane0[ne]];
        if (c >= bc - RM_CONT_EPS || c >= RM_GOOD_CONT) n_acc += 1;
    }
    int32_t pick = n_acc >= 2 ? (trip & 1) : 0;     /* even trips take the first, odd trips the second of them */
    for (int32_t l = ls; l < ls + lc; ++l) {
        if (sc->link_to_edge[l] != ne) continue;
        float c = cn[sc->link_dest_lane[l] - sc->edge_lane0[ne]];
        if (c >= bc - RM_CONT_EPS || c >= RM_GOOD_CONT) { if (pick == 0) return l; pick -= 1; }
    }
    return -1;
}
/* strategic lane-change need of the vehicle in slot s (on a normal lane): 0 when its lane is as good as any of the edge
 * or the need is still far away, else the direction (+1 left, -1 right) of the nearest best lane.  *rem = how far it can
 * still drive on its lane [SUMO-K LC2013: a change is due when rem < lookahead * number of lanes to cross, lookahead =
 * 10 s at the current speed (at least 5 m/s) + 10 m; `extra` = 2 when asking whether a lane is good enough to move INTO for
 * speed gain: LC2013 leaves the best lanes only if it can stay away for (lanes + 2) look-aheads] */
static int32_t strategic_dir_at(const orc_env *e, int32_t route, int32_t cursor, int32_t kk, int32_t n, float x, float v, int extra, float *rem) {
    const orc_scenario *sc = e->sc;
    /* [SUMO-K LC2013 _wantsChange: usableDist = currentDist - posOnLane - best.occupation * JAM_FACTOR] the vehicles standing
     * or driving on the lane I have to get to shorten the distance I can still use: a vehicle that needs the neighbouring lane
     * joins the queue there at its tail instead of driving past it.  Occupation = vehicles on that lane now * occ_unit */
    const float *cn = sc->route_cont + (size_t)(sc->route_start[route] + cursor) * sc->kmax;
    float best = 0.0f;
    for (int32_t j = 0; j < n; ++j) if (cn[j] > best) best = cn[j];
    *rem = cn[kk] - x;
    if (cn[kk] >= best - RM_CONT_EPS) return 0;
    int32_t dl = 1000, dr = 1000;
    for (int32_t j = kk + 1; j < n; ++j) if (cn[j] >= best - RM_CONT_EPS) { dl = j - kk; break; }
    for (int32_t j = kk - 1; j >= 0; --j) if (cn[j] >= best - RM_CONT_EPS) { dr = kk - j; break; }
    int32_t off = (dr <= dl ? dr : dl) + extra;
    float la = (v > RM_LOOK_MIN_SPEED ? v : RM_LOOK_MIN_SPEED) * RM_LOOK_TIME + RM_LOOK_BASE;
    int32_t bl = sc->edge_lane0[sc->route_edge[sc->route_start[route] + cursor]] + ((dr <= dl) ? kk - dr : kk + dl);
    if (*rem - (float)e->lane_cnt[bl] * (e->occ_unit * RM_OCC_FACTOR) >= la * (float)off) return 0;
    return (dr <= dl) ? -1 : +1;
}
static inline int32_t tls_state(const orc_env *e, int32_t link) {
    const orc_scenario *sc = e->sc;
    int32_t s = sc->link_tls[link];
    if (s < 0) return TLS_G;
    if (e->p.fixed_program)
        return sc->fix_states[sc->fix_state_off[s] + e->phase[s] * sc->tls_nlinks[s] + sc->link_tls_pos[link]];
    return sc->tls_states[sc->tls_state_off[s] + e->phase[s] * sc->tls_nlinks[s] + sc->link_tls_pos[link]];
}
/* departure lane [SUMO-K departLane default "first"]: the right-most lane of the first edge the vehicle may use */
static inline int32_t depart_lane(const orc_scenario *sc, int32_t route) {
    return sc->edge_lane0[sc->route_edge[sc->route_start[route]]];
}
static void build_lists(orc_env *e) {
    const orc_scenario *sc = e->sc;
    for (int32_t l = 0; l < sc->n_lanes; ++l) { e->lane_head[l] = NIL; e->lane_cnt[l] = 0; }
    for (int32_t s = 0; s < e->hw; ++s) {
        if (e->lane[s] >= LANE_PENDING) continue;
        e->next_in_lane[s] = e->lane_head[e->lane[s]];
        e->lane_head[e->lane[s]] = s;
        e->lane_cnt[e->lane[s]] += 1;
    }
}
/* rear-most vehicle of a lane (min pos, ties -> larger trip index) if its front is within `win` metres of the lane start.
 * Vehicles further away cannot influence anybody who looks `win` metres ahead; the cut makes the search a bounded scan. */
static int32_t rearmost(const orc_env *e, int32_t lane, float win) {
    int32_t best = NIL, bk = 0;
    for (int32_t s = e->lane_head[lane]; s != NIL; s = e->next_in_lane[s]) {
        int32_t k = trip_of_slot(e, s);
        if (best == NIL || e->pos[s] < e->pos[best] || (e->pos[s] == e->pos[best] && k > bk)) { best = s; bk = k; }
    }
    if (best != NIL && e->pos[best] > win) return NIL;
    return best;
}
/* nearest vehicle ahead / behind of (pos,k) on a lane, at most `win` metres away (front to front) */
static void neighbours(const orc_env *e, int32_t lane, float pos, int32_t k, int32_t self, float win, int32_t *lead, int32_t *foll) {
    int32_t L = NIL, F = NIL, Lk = 0, Fk = 0;
    for (int32_t s = e->lane_head[lane]; s != NIL; s = e->next_in_lane[s]) {
        if (s == self) continue;
        int32_t ks = trip_of_slot(e, s);
        if (ahead_of(e->pos[s], ks, pos, k)) {
            if (L == NIL || ahead_of(e->pos[L], Lk, e->pos[s], ks)) { L = s; Lk = ks; }
        } else {
            if (F == NIL || ahead_of(e->pos[s], ks, e->pos[F], Fk)) { F = s; Fk = ks; }
        }
    }
    if (L != NIL && e->pos[L] - pos > win) L = NIL;
    if (F != NIL && pos - e->pos[F] > win) F = NIL;
    *lead = L; *foll = F;
}

/* ------------------------------------------------------------------ lifecycle */
#define ALLOC(p, n) p = calloc((size_t)(n) > 0 ? (size_t)(n) : 1, sizeof(*(p)))
orc_env *orc_create(const orc_scenario *sc, const orc_params *p, int32_t env_index) {
    orc_env *e = calloc(1, sizeof(orc_env));
    e->sc = sc; e->p = *p; e->env_index = env_index;
    int32_t C = sc->capacity, S = sc->n_signals, O = sc->n_obs;
    ALLOC(e->lane, C); ALLOC(e->cursor, C); ALLOC(e->sumo_wait, C); ALLOC(e->resco_wait, C); ALLOC(e->depart, C);
    ALLOC(e->owner, C); ALLOC(e->pos, C); ALLOC(e->speed, C); ALLOC(e->accel, C); ALLOC(e->time_loss, C);
    ALLOC(e->vnext, C); ALLOC(e->lc_target, C); ALLOC(e->trip, C); ALLOC(e->dbg_reason, C); ALLOC(e->dbg_block, C); ALLOC(e->wtot, C);
    if (p->trip_log) ALLOC(e->trip_log, (size_t)sc->n_trips * 4);
    ALLOC(e->lane_head, sc->n_lanes); ALLOC(e->next_in_lane, C); ALLOC(e->link_arr, sc->n_links);
    ALLOC(e->lane_ins, sc->n_lanes); ALLOC(e->lane_cnt, sc->n_lanes);
    ALLOC(e->dep_next, sc->n_lanes); ALLOC(e->dep_first, sc->n_lanes); ALLOC(e->trip_next, sc->n_trips); ALLOC(e->coop, C); ALLOC(e->coop_lead, C); ALLOC(e->free_before, C); ALLOC(e->coop_lead_trip, C);
    {   /* the trips of one departure lane form a FIFO in trip (= departure time) order */
        for (int32_t l = 0; l < sc->n_lanes; ++l) e->dep_first[l] = -1;
        for (int32_t k = sc->n_trips - 1; k >= 0; --k) {
            int32_t dl = depart_lane(sc, sc->trip_route[k]);
            e->trip_next[k] = e->dep_first[dl];
            e->dep_first[dl] = k;
        }
    }
    ALLOC(e->phase, S); ALLOC(e->left, S); ALLOC(e->next_phase, S);
    ALLOC(e->lane_agg, O * 5); ALLOC(e->drq_norm, O * 5); ALLOC(e->wait, S); ALLOC(e->wait_norm, S);
    ALLOC(e->agg_q, O); ALLOC(e->agg_a, O); ALLOC(e->agg_w, O); ALLOC(e->agg_m, O); ALLOC(e->agg_s, O);
    ALLOC(e->out_phase, S); ALLOC(e->mplight, S * 13); ALLOC(e->wave, S * 12); ALLOC(e->pressure, S);
    ALLOC(e->queue_sum, S); ALLOC(e->queue_max, S);
    ALLOC(e->sig_arr, S); ALLOC(e->sig_dep, S); ALLOC(e->out_arr, S); ALLOC(e->out_dep, S); ALLOC(e->mplight_full, S * 49); ALLOC(e->lane_arr, sc->n_obs);
    {   /* the most common vehicle type (ties: the lower index) */
        int32_t best = 0, bestn = -1;
        for (int32_t v = 0; v < sc->n_vtypes; ++v) {
            int32_t c = 0;
            for (int32_t k = 0; k < sc->n_trips; ++k) if (sc->trip_vtype[k] == v) c += 1;
            if (c > bestn) { bestn = c; best = v; }
        }
        e->occ_unit = sc->vtype_params[best * VT_COLS + VT_LENGTH] + sc->vtype_params[best * VT_COLS + VT_MINGAP];
    }
    e->maxlen = 0.0f;
    for (int32_t v = 0; v < sc->n_vtypes; ++v) if (sc->vtype_params[v * VT_COLS + VT_LENGTH] > e->maxlen) e->maxlen = sc->vtype_params[v * VT_COLS + VT_LENGTH];
    orc_reset(e);
    return e;
}
void orc_destroy(orc_env *e) {
    if (!e) return;
    free(e->lane); free(e->cursor); free(e->sumo_wait); free(e->resco_wait); free(e->depart); free(e->owner);
    free(e->pos); free(e->speed); free(e->accel); free(e->time_loss); free(e->vnext); free(e->lc_target); free(e->trip); free(e->dbg_reason); free(e->dbg_block); free(e->wtot); free(e->trip_log);
    free(e->lane_head); free(e->next_in_lane); free(e->link_arr); free(e->lane_ins); free(e->lane_cnt);
    free(e->dep_next); free(e->dep_first); free(e->trip_next); free(e->coop); free(e->coop_lead); free(e->free_before); free(e->coop_lead_trip);
    free(e->phase); free(e->left); free(e->next_phase);
    free(e->lane_agg); free(e->drq_norm); free(e->wait); free(e->wait_norm);
    free(e->agg_q); free(e->agg_a); free(e->agg_w); free(e->agg_m); free(e->agg_s);
    free(e->sig_arr); free(e->sig_dep); free(e->out_arr); free(e->out_dep); free(e->mplight_full); free(e->lane_arr);
    free(e->out_phase); free(e->mplight); free(e->wave); free(e->pressure); free(e->queue_sum); free(e->queue_max);
    free(e);
}
void orc_reset(orc_env *e) {
    const orc_scenario *sc = e->sc;
    e->t = 0; e->n_inserted = 0; e->n_active = 0; e->hw = 0;
    for (int32_t l = 0; l < sc->n_lanes; ++l) e->dep_next[l] = e->dep_first[l];
    for (int32_t s = 0; s < sc->capacity; ++s) { e->coop[s] = -1; e->coop_lead[s] = -1; }
    for (int32_t s = 0; s < sc->capacity; ++s) {
        e->lane[s] = LANE_NONE; e->owner[s] = OWNER_NONE; e->resco_wait[s] = 0; e->sumo_wait[s] = 0; e->trip[s] = -1;
        e->pos[s] = 0; e->speed[s] = 0; e->accel[s] = 0; e->time_loss[s] = 0; e->cursor[s] = 0; e->depart[s] = 0;
    }
    for (int32_t l = 0; l < sc->n_links; ++l) e->link_arr[l] = ARR_NONE;
    for (int32_t l = 0; l < sc->n_lanes; ++l) e->lane_ins[l] = 0x7FFFFFFF;
    for (int32_t s = 0; s < sc->n_signals; ++s) {
        if (e->p.fixed_program) {
            e->phase[s] = sc->fix_init_phase[s];
            e->left[s] = sc->fix_init_left[s];
        } else {
            /* [SUMO-K] setProgramLogic (traffic_signal.py:96-100) restarts the current index with its full duration */
            e->phase[s] = sc->tls_init_phase[s];
            e->left[s] = sc->tls_dur[sc->tls_dur_off[s] + e->phase[s]];
        }
        e->next_phase[s] = 0;       /* Signal.__init__: self.next_phase = 0 (traffic_signal.py:32) */
        e->sig_arr[s] = 0; e->sig_dep[s] = 0;
    }
    memset(e->stats, 0, sizeof(e->stats));
    if (e->trip_log) memset(e->trip_log, 0, (size_t)sc->n_trips * 4 * sizeof(int32_t));
    build_lists(e);
}
int32_t orc_time(const orc_env *e) { return e->t; }
int32_t orc_get_phase(const orc_env *e, int32_t sig) { return e->phase[sig]; }
void orc_set_phase(orc_env *e, int32_t sig, int32_t ph) {
    const orc_scenario *sc = e->sc;
    if (ph < 0 || ph >= sc->tls_nphase[sig]) return;
    e->phase[sig] = ph;
    /* trafficlight.setPhase [SUMO-K, unpinned; see rs_params.tls_hold]: the phase runs for its programme duration and the programme then
     * continues with the next index (default); with tls_hold the phase stays until the next setPhase */
    e->left[sig] = e->p.tls_hold ? RM_TLS_HOLD_TICKS : sc->tls_dur[sc->tls_dur_off[sig] + ph];
}

/* ------------------------------------------------------------------ the tick */
static void tls_events(orc_env *e) {
    const orc_scenario *sc = e->sc;
    for (int32_t s = 0; s < sc->n_signals; ++s) {
        const int32_t *dur = e->p.fixed_program ? sc->fix_dur + sc->fix_dur_off[s] : sc->tls_dur + sc->tls_dur_off[s];
        int32_t P = e->p.fixed_program ? sc->fix_nphase[s] : sc->tls_nphase[s];
        if (e->left[s] == 0) {
            e->phase[s] = (e->phase[s] + 1) % P;
            e->left[s] = dur[e->phase[s]];
        }
        e->left[s] -= 1;
    }
}

/* [SUMO-K] MSInsertionControl (emitVehicles runs at the end of a simulation step, after the lane changes): every
 * departure lane keeps its own backlog; its oldest trip is inserted (departPos "base", departSpeed 0) as soon as it has
 * departed and the space behind the rear-most vehicle suffices AFTER this tick's move of the vehicles that are on the lane now
 * (their planned speeds are known; a vehicle that enters the lane in this very tick is not seen). */
static void insertion_check(orc_env *e) {
    const orc_scenario *sc = e->sc;
    for (int32_t dl = 0; dl < sc->n_lanes; ++dl) {
        e->lane_ins[dl] = -1;
        int32_t k = e->dep_next[dl];
        if (k < 0 || sc->trip_depart[k] > e->t) continue;
        const float *vt = vt_of(e, k);
        float mypos = vt[VT_LENGTH] < sc->lane_len[dl] ? vt[VT_LENGTH] : sc->lane_len[dl];
        int ok = 1;
        for (int32_t o = e->lane_head[dl]; o != NIL; o = e->next_in_lane[o]) {
            const float *vo = vt_of(e, trip_of_slot(e, o));
            float back = (e->pos[o] + e->vnext[o]) - vo[VT_LENGTH];       /* where it will be after this tick's move */
            if (back - mypos - vt[VT_MINGAP] < 0.0f) ok = 0;
        }
        if (ok) e->lane_ins[dl] = k;
    }
    /* the winners take the slots that are free NOW (before this tick's arrivals free more), lower lane index first */
    e->room_ins = sc->capacity - e->n_active;
    for (int32_t s = 0; s < sc->capacity; ++s) e->free_before[s] = e->lane[s] == LANE_NONE;
}
static void insertion_apply(orc_env *e) {
    const orc_scenario *sc = e->sc;
    int32_t C = sc->capacity;
    int32_t room = e->room_ins;             /* the network holds at most `capacity` vehicles */
    /* stats[10]: insertions refused because every slot was taken (the trip stays in its backlog): winners beyond the free capacity */
    {
        int32_t winners = 0;
        for (int32_t dl = 0; dl < sc->n_lanes; ++dl) if (e->lane_ins[dl] >= 0) winners += 1;
        if (winners > room) e->stats[10] += winners - (room > 0 ? room : 0);
    }
    for (int32_t dl = 0; dl < sc->n_lanes && room > 0; ++dl) {      /* lower lane index first when the network is full */
        int32_t k = e->lane_ins[dl];
        if (k < 0) continue;
        int32_t s = 0;
        while (s < C && !e->free_before[s]) s += 1;                 /* lowest slot that was free before this tick (the index has no meaning) */
        if (s >= C) break;
        e->free_before[s] = 0;
        const float *vt = vt_of(e, k);
        e->trip[s] = k; e->lane[s] = (uint16_t)dl;
        e->pos[s] = vt[VT_LENGTH] < sc->lane_len[dl] ? vt[VT_LENGTH] : sc->lane_len[dl];
        e->speed[s] = 0; e->accel[s] = 0; e->time_loss[s] = 0; e->cursor[s] = 0;
        e->sumo_wait[s] = 0; e->resco_wait[s] = 0; e->owner[s] = OWNER_NONE; e->depart[s] = (uint16_t)(e->t + 1); e->wtot[s] = 0;
        e->coop[s] = -1; e->coop_lead[s] = -1;
        if (s + 1 > e->hw) e->hw = s + 1;
        e->dep_next[dl] = e->trip_next[k];
        e->n_inserted += 1; e->n_active += 1; room -= 1;
        e->stats[0] += 1;
        e->stats[3] += e->t - sc->trip_depart[k];
    }
}

static void register_approaches(orc_env *e) {
    const orc_scenario *sc = e->sc;
    for (int32_t s = 0; s < e->hw; ++s) {
        int32_t k = e->trip[s];
        if (e->lane[s] >= LANE_PENDING) continue;
        int32_t lane = e->lane[s];
        int32_t link = choose_link(e, lane, sc->trip_route[k], e->cursor[s], k);
        if (link < 0) continue;
        const float *vt = vt_of(e, k);
        float dist = sc->lane_len[lane] - e->pos[s];
        float v = e->speed[s];
        int32_t st = tls_state(e, link);
        if (v <= HALT_SPEED) continue;          /* a standing vehicle is not "approaching" (SUMO willPass=false) */
        if (st == TLS_R) continue;
        if (st == TLS_Y && dist >= orc_brake_gap(v, vt[VT_DECEL])) continue;
        float ta = dist / (v > 1.0f ? v : 1.0f);
        int32_t q = ta * 10.0f >= 65000.0f ? 65000 : (int32_t)(ta * 10.0f);
        if (q < e->link_arr[link]) e->link_arr[link] = q;
    }
}

/* a vehicle moving on a foe's junction lane blocks; a standing one (spill-back) is driven around */
static int lane_has_mover(const orc_env *e, int32_t lane) {
    for (int32_t s = e->lane_head[lane]; s != NIL; s = e->next_in_lane[s])
        if (e->speed[s] > HALT_SPEED) return 1;
    return 0;
}
static int foe_blocked(const orc_env *e, int32_t link) {
    const orc_scenario *sc = e->sc;
    int32_t fs = sc->link_foe_start[link], fc = sc->link_foe_cnt[link];
    for (int32_t i = fs; i < fs + fc; ++i) {
        int32_t f = sc->foe_link[i];
        if (sc->link_tls[f] >= 0 && tls_state(e, f) == TLS_R) continue;
        if (e->link_arr[f] < RM_FOE_GAP_Q) return 1;
        if (sc->link_via1[f] >= 0 && lane_has_mover(e, sc->link_via1[f])) return 1;
        if (sc->link_via2[f] >= 0 && lane_has_mover(e, sc->link_via2[f])) return 1;
    }
    return 0;
}

static void plan(orc_env *e) {
    const orc_scenario *sc = e->sc;
    for (int32_t s = 0; s < e->hw; ++s) {
        int32_t k = e->trip[s];
        if (e->lane[s] >= LANE_PENDING) continue;
        const float *vt = vt_of(e, k);
        float a = vt[VT_ACCEL], b = vt[VT_DECEL], tau = vt[VT_TAU], mingap = vt[VT_MINGAP];
        int32_t lane = e->lane[s], route = sc->trip_route[k];
        int32_t cursor = e->cursor[s];
        float v = e->speed[s], x = e->pos[s];
        float sf = speed_factor(e, k);
        float vfree = v + a;
        float vl = sc->lane_vmax[lane] * sf;
        if (vl < vfree) vfree = vl;
        if (vt[VT_MAXSPEED] < vfree) vfree = vt[VT_MAXSPEED];
        float vsafe = BIGF;
        e->dbg_reason[s] = 0; e->dbg_block[s] = -1;
        /* leader on my own lane (one further away than I look ahead cannot matter) */
        float look = orc_brake_gap(vfree, b) + vfree * tau + mingap + 1.0f;
        int32_t lead, foll;
        neighbours(e, lane, x, k, s, look + e->maxlen, &lead, &foll);
        int found = 0;
        if (lead != NIL) {
            const float *vo = vt_of(e, trip_of_slot(e, lead));
            float gap = e->pos[lead] - vo[VT_LENGTH] - x - mingap;
            vsafe = orc_follow_speed(gap, e->speed[lead], b, vo[VT_DECEL], tau);
            found = 1;
            e->dbg_reason[s] = 1; e->dbg_block[s] = lead;
        }
        /* cooperation [SUMO-K LC2013 informFollower]: a vehicle on the neighbouring lane that has to get into my lane and
         * found no gap asked me (in the last lane-change phase) to let it in: I follow it as if it were already there,
         * braking no harder than comfortably */
        if (e->coop[s] >= 0) {
            int32_t X = (int32_t)(e->coop[s] & 0xFFFFFFFF), kx = (int32_t)(e->coop[s] >> 32);
            e->coop[s] = -1;
            if (e->trip[X] == kx && e->lane[X] < LANE_PENDING && !sc->lane_internal[e->lane[X]] && !sc->lane_internal[lane] &&
                sc->lane_edge[e->lane[X]] == sc->lane_edge[lane]) {
                const float *vo = vt_of(e, kx);
                float backx = e->pos[X] - vo[VT_LENGTH];
                if (backx >= x) {
                    float vs = orc_follow_speed(backx - x - mingap, e->speed[X], b, vo[VT_DECEL], tau);
                    float vc = v - b; if (vc < 0.0f) vc = 0.0f;
                    if (vs < vc) vs = vc;
                    if (vs < vsafe) { vsafe = vs; e->dbg_reason[s] = 8; e->dbg_block[s] = X; }
                }
            }
        }
        /* ... and when I am the one who has to change: fall in behind the vehicle ahead of me on the target lane [informLeader] */
        if (e->coop_lead[s] >= 0) {
            int32_t X = e->coop_lead[s], kx = e->coop_lead_trip[s];
            e->coop_lead[s] = -1;
            if (e->trip[X] == kx && e->lane[X] < LANE_PENDING && !sc->lane_internal[e->lane[X]] && !sc->lane_internal[lane] &&
                sc->lane_edge[e->lane[X]] == sc->lane_edge[lane]) {
                const float *vo = vt_of(e, kx);
                float backx = e->pos[X] - vo[VT_LENGTH];
                if (e->pos[X] >= x) {
                    float g = backx - x - mingap;
                    float vs = orc_follow_speed(g > 0.0f ? g : 0.0f, e->speed[X], b, vo[VT_DECEL], tau);
                    float vc = v - b; if (vc < 0.0f) vc = 0.0f;
                    if (vs < vc) vs = vc;
                    if (vs < vsafe) { vsafe = vs; e->dbg_reason[s] = 9; e->dbg_block[s] = X; }
                }
            }
        }
        /* look ahead along my path */
        float seen = sc->lane_len[lane] - x;
        int32_t cur = lane, cur_cursor = cursor;
        int32_t rn = sc->route_start[route + 1] - sc->route_start[route];
        for (int hop = 0; hop < MAX_HOPS && !found && seen < look; ++hop) {
            if (!sc->lane_internal[cur] && cur_cursor + 1 >= rn) break;     /* my last edge: free run to its end */
            int32_t link = choose_link(e, cur, route, cur_cursor, k);
            if (link < 0) {     /* wrong lane for my route: wait at the end for a lane change */
                float g = seen - STOP_OFFSET;
                float vs = orc_stop_speed(g > 0.0f ? g : 0.0f, b, tau);
                if (vs < vsafe) { vsafe = vs; e->dbg_reason[s] = 2; e->dbg_block[s] = cur; }
                break;
            }
            int32_t st = tls_state(e, link);
            int stop_here = 0;
            if (sc->link_tls[link] >= 0 && (st == TLS_R || st == TLS_Y)) {
                if (seen >= orc_brake_gap(v, b)) { stop_here = 1; e->dbg_reason[s] = 3; e->dbg_block[s] = link; }
            }
            if (!stop_here && !sc->link_cont[link] &&
                (sc->link_minor[link] || (sc->link_tls[link] >= 0 && st == TLS_g))) {
                /* [SUMO-K] MSVehicle::processLinkApproaches: a minor link is approached as if one had to stop until the
                 * foe lanes can be seen (foe visibility distance 4.5 m) */
                if (seen > RM_VIS_DIST) { stop_here = 1; e->dbg_reason[s] = 7; e->dbg_block[s] = link; }
                else if (sc->link_foe_cnt[link] > 0 && foe_blocked(e, link)) { stop_here = 1; e->dbg_reason[s] = 4; e->dbg_block[s] = link; }
            }
            if (stop_here) {
                float g = seen - STOP_OFFSET;
                float vs = orc_stop_speed(g > 0.0f ? g : 0.0f, b, tau);
                if (vs < vsafe) vsafe = vs;
                break;
            }
            int32_t nl = sc->link_to_lane[link];
            {   /* slow down in time for a lower speed limit on the next lane */
                float vnl = sc->lane_vmax[nl] * sf;
                if (vnl < vfree) {
                    float vs = orc_free_speed(seen, vnl, b);
                    if (vs < vsafe) { vsafe = vs; e->dbg_reason[s] = 6; e->dbg_block[s] = nl; }
                }
            }
            int32_t o = rearmost(e, nl, look - seen + e->maxlen);
            if (o != NIL) {
                const float *vo = vt_of(e, trip_of_slot(e, o));
                float gap = seen + e->pos[o] - vo[VT_LENGTH] - mingap;
                float vs = orc_follow_speed(gap, e->speed[o], b, vo[VT_DECEL], tau);
                if (vs < vsafe) { vsafe = vs; e->dbg_reason[s] = 5; e->dbg_block[s] = o; }
                found = 1;
                break;
            }
            if (!sc->lane_internal[cur]) cur_cursor += 1;
            seen += sc->lane_len[nl];
            cur = nl;
        }
        /* MSCFModel::finalizeSpeed + Krauss dawdle2 [SUMO-K] */
        float vmin_n = v - b; if (vmin_n < 0.0f) vmin_n = 0.0f;
        float vmin_e = v - vt[VT_EMERGENCY]; if (vmin_e < 0.0f) vmin_e = 0.0f;
        float lo = vsafe > vmin_e ? vsafe : vmin_e;
        float vmin = vmin_n < lo ? vmin_n : lo;
        float vmax = vfree < vsafe ? vfree : vsafe;
        if (vmax < vmin) vmax = vmin;
        float sigma = e->p.sigma >= 0.0f ? e->p.sigma : vt[VT_SIGMA];
        float vd = vmax;
        if (sigma > 0.0f) {
            float r = u01(orc_hash(e->p.seed, (uint32_t)e->env_index, (uint32_t)k, (uint32_t)e->t, 0u));
            if (vd < a) vd -= sigma * vd * r; else vd -= sigma * a * r;
            if (vd < 0.0f) vd = 0.0f;
        }
        e->vnext[s] = vd > vmin ? vd : vmin;
    }
}

static void move(orc_env *e) {
    const orc_scenario *sc = e->sc;
    for (int32_t s = 0; s < e->hw; ++s) {
        int32_t k = e->trip[s];
        if (e->lane[s] >= LANE_PENDING) continue;
        /* clear my approach registration (the table is all-ARR_NONE between ticks) */
        int32_t lk = choose_link(e, e->lane[s], sc->trip_route[k], e->cursor[s], k);
        if (lk >= 0) e->link_arr[lk] = ARR_NONE;
    }
    int32_t active = 0;
    for (int32_t s = 0; s < e->hw; ++s) {
        int32_t k = e->trip[s];
        if (e->lane[s] >= LANE_PENDING) continue;
        int32_t route = sc->trip_route[k];
        int32_t rn = sc->route_start[route + 1] - sc->route_start[route];
        float vn = e->vnext[s];
        float sf = speed_factor(e, k);
        float vref = sc->lane_vmax[e->lane[s]] * sf;
        e->accel[s] = vn - e->speed[s];
        e->speed[s] = vn;
        if (vn <= HALT_SPEED) { if (e->sumo_wait[s] < 65535) e->sumo_wait[s] += 1; e->stats[4] += 1; if (e->p.trip_log && e->wtot[s] < 65535) e->wtot[s] += 1; }
        else e->sumo_wait[s] = 0;
        if (vref > 0.0f && vn < vref) e->time_loss[s] += (vref - vn) / vref;
        float x = e->pos[s] + vn;
        int32_t lane = e->lc_target[s] >= 0 ? e->lc_target[s] : e->lane[s], cursor = e->cursor[s];       /* sideways first, then forward */
        int arrived = 0;
        for (int it = 0; it < 16; ++it) {
            float len = sc->lane_len[lane];
            if (!(x > len)) break;
            if (!sc->lane_internal[lane] && cursor + 1 >= rn) { arrived = 1; break; }
            int32_t link = choose_link(e, lane, route, cursor, k);
            if (link < 0) { x = len; break; }
            x -= len;
            if (!sc->lane_internal[lane]) cursor += 1;
            lane = sc->link_to_lane[link];
        }
        if (arrived) {
            if (e->owner[s] != OWNER_NONE) e->sig_dep[e->owner[s]] += 1;    /* Signal.departures of its last observer (traffic_signal.py:226-232) */
            e->lane[s] = LANE_NONE; e->owner[s] = OWNER_NONE; e->resco_wait[s] = 0; e->trip[s] = -1;
            e->n_active -= 1;
            e->stats[1] += 1;
            e->stats[2] += e->t + 1 - e->depart[s];
            e->stats[5] += (int64_t)(e->time_loss[s] * 1024.0f + 0.5f);
            if (e->trip_log) {
                int32_t *r = e->trip_log + (size_t)k * 4;
                r[0] = e->depart[s]; r[1] = e->t + 1; r[2] = (int32_t)(e->time_loss[s] * 1024.0f + 0.5f); r[3] = e->wtot[s];
            }
        } else {
            e->lane[s] = (uint16_t)lane; e->cursor[s] = (uint16_t)cursor; e->pos[s] = x;
            active += 1;
        }
    }
    e->stats[8] += active;
    while (e->hw > 0 && e->lane[e->hw - 1] == LANE_NONE) e->hw -= 1;
}

/* Mutual block (own rule; SUMO resolves the same situation with its sublane / cooperative machinery or teleports, which this
 * model does not have): two stationary vehicles stand side by side near the end of their lanes, each in the lane the other
 * one needs.  Neither can ever find a gap, so they trade places.  swap_dir: the strategic direction of such a vehicle (0:
 * it is not one). */
static int32_t swap_dir(const orc_env *e, int32_t s) {
    const orc_scenario *sc = e->sc;
    if (e->lane[s] >= LANE_PENDING) return 0;
    int32_t lane = e->lane[s];
    if (sc->lane_internal[lane] || e->speed[s] > HALT_SPEED || e->sumo_wait[s] < RM_SWAP_WAIT) return 0;
    int32_t ed = sc->lane_edge[lane], n = sc->edge_nlanes[ed], kk = lane - sc->edge_lane0[ed];
    if (n < 2) return 0;
    float rem;
    int32_t d = strategic_dir_at(e, sc->trip_route[e->trip[s]], e->cursor[s], kk, n, e->pos[s], e->speed[s], 0, &rem);
    if (d == 0 || rem > RM_URGENT_DIST) return 0;
    return d;
}
/* the vehicle on lane tl whose body overlaps mine lengthwise (the nearer one ahead first), NIL: none */
static int32_t overlapping(const orc_env *e, int32_t s, int32_t tl) {
    int32_t lead, foll;
    int32_t k = e->trip[s];
    neighbours(e, tl, e->pos[s], k, s, RM_NB_WINDOW, &lead, &foll);
    if (lead != NIL && e->pos[lead] - vt_of(e, trip_of_slot(e, lead))[VT_LENGTH] - e->pos[s] < 0.0f) return lead;
    if (foll != NIL && e->pos[s] - vt_of(e, k)[VT_LENGTH] - e->pos[foll] < 0.0f) return foll;
    return NIL;
}
static void lane_change(orc_env *e) {
    const orc_scenario *sc = e->sc;
    int32_t dir_allowed = (e->t & 1) ? -1 : +1;
    for (int32_t s = 0; s < e->hw; ++s) {
        int32_t k = e->trip[s];
        e->lc_target[s] = -1;
        if (e->lane[s] >= LANE_PENDING) continue;
        int32_t lane = e->lane[s];
        if (sc->lane_internal[lane]) continue;
        int32_t ed = sc->lane_edge[lane];
        int32_t n = sc->edge_nlanes[ed];
        if (n < 2) continue;
        int32_t kk = lane - sc->edge_lane0[ed];
        int32_t route = sc->trip_route[k];
        const float *vt = vt_of(e, k);
        float x = e->pos[s], v = e->speed[s];
        int want = 0, dir = dir_allowed;
        float rem;
        int32_t sdir = strategic_dir_at(e, route, e->cursor[s], kk, n, x, v, 0, &rem);
        if (sdir != 0) {
            /* strategic: head for the nearest lane that continues my route.  The target lane is examined on every tick
             * (a blocked vehicle asks for cooperation), the change itself happens on the ticks of its direction */
            dir = sdir;
                        want = 2;
        }
        int32_t tk = kk + dir;
        if (tk < 0 || tk >= n) continue;
        int32_t tl = sc->edge_lane0[ed] + tk;
        int32_t lead_c, foll_c, lead_t, foll_t;
        neighbours(e, tl, x, k, s, RM_NB_WINDOW, &lead_t, &foll_t);
        float rem_t;
        if (!want && ((((uint32_t)e->t >> 1) + (uint32_t)k) & 3u) == 0u && strategic_dir_at(e, route, e->cursor[s], tk, n, x, v, RM_SG_EXTRA_LANES, &rem_t) == 0) {
            /* speed gain between equally good lanes: more room ahead on the neighbour.  A vehicle reconsiders
             * only on one pair of ticks (one left, one right chance) out of four (LC2013 needs several
             * seconds of accumulated speed-gain incentive before it acts [SUMO-K]) */
            neighbours(e, lane, x, k, s, RM_NB_WINDOW, &lead_c, &foll_c);
            if (lead_c != NIL) {
                const float *vo = vt_of(e, trip_of_slot(e, lead_c));
                float gcur = e->pos[lead_c] - vo[VT_LENGTH] - x;
                float gtgt = BIGF;
                if (lead_t != NIL) {
                    const float *vq = vt_of(e, trip_of_slot(e, lead_t));
                    gtgt = e->pos[lead_t] - vq[VT_LENGTH] - x;
                }
                if (gcur < v * 3.0f + 15.0f && gtgt > gcur + RM_SG_ADVANTAGE) want = 1;
            }
        }
        if (!want) continue;
        /* urgent = strategic change close to the end of the lane: accept tighter gaps (followers may have to
         * brake with their emergency deceleration), otherwise dense queues would never let anybody in */
        int urgent = want == 2 && rem <= RM_URGENT_DIST;
        int safe = 1;
        if (lead_t != NIL) {
            const float *vo = vt_of(e, trip_of_slot(e, lead_t));
            float gap = e->pos[lead_t] - vo[VT_LENGTH] - x - (urgent ? 0.0f : vt[VT_MINGAP]);
            float dec = urgent ? vt[VT_EMERGENCY] : vt[VT_DECEL];
            float vb = v - dec; if (vb < 0.0f) vb = 0.0f;
            if (gap < 0.0f || vb > orc_follow_speed(gap, e->speed[lead_t], vt[VT_DECEL], vo[VT_DECEL], vt[VT_TAU])) safe = 0;
        }
        if (safe && foll_t != NIL) {
            const float *vo = vt_of(e, trip_of_slot(e, foll_t));
            float gap = x - vt[VT_LENGTH] - e->pos[foll_t] - (urgent ? 0.0f : vo[VT_MINGAP]);
            float dec = urgent ? vo[VT_EMERGENCY] : vo[VT_DECEL];
            float vb = e->speed[foll_t] - dec; if (vb < 0.0f) vb = 0.0f;
            if (gap < 0.0f || vb > orc_follow_speed(gap, v, vo[VT_DECEL], vt[VT_DECEL], vo[VT_TAU])) safe = 0;
        }
        if (safe) {
            /* (a vehicle that leaves its lane in this tick changes lanes on the next edge, if at all: its plan looked at the
             *  links of the lane it is on) */
            if (dir == dir_allowed && !(x + e->vnext[s] > sc->lane_len[lane])) e->lc_target[s] = tl;
            continue;
        }
        if (want == 2 && lead_t != NIL) { e->coop_lead[s] = lead_t; e->coop_lead_trip[s] = trip_of_slot(e, lead_t); }
        if (want == 2) {
            /* blocked: ask the nearest vehicle of the target lane that is completely behind me to let me in */
            float back = x - vt[VT_LENGTH];
            int32_t R = NIL, Rk = 0;
            for (int32_t o = e->lane_head[tl]; o != NIL; o = e->next_in_lane[o]) {
                int32_t ko = trip_of_slot(e, o);
                if (e->pos[o] > back || back - e->pos[o] > RM_COOP_RANGE) continue;
                if (R == NIL || ahead_of(e->pos[o], ko, e->pos[R], Rk)) { R = o; Rk = ko; }
            }
            if (R != NIL) {
                int64_t key = ((int64_t)k << 32) | (int64_t)s;
                if (e->coop[R] < 0 || key < e->coop[R]) e->coop[R] = key;
            }
        }
    }
    /* mutual blocks: the test is symmetric, so both vehicles reach the same verdict (whatever this tick's direction is) */
    for (int32_t s = 0; s < e->hw && e->t % RM_SWAP_EVERY == 0; ++s) {
        int32_t d = swap_dir(e, s);
        if (d == 0 || e->lc_target[s] >= 0) continue;
        int32_t lane = e->lane[s], tl = lane + d;
        int32_t b = overlapping(e, s, tl);
        if (b == NIL || swap_dir(e, b) != -d) continue;
        if (overlapping(e, b, lane) != s) continue;
        e->lc_target[s] = tl;
    }
}
/* One tick.  All decisions of a tick -- speeds, lane changes, insertions -- are taken on the state at its beginning and
 * executed together (SUMO executes the moves before it decides the lane changes; with both on one state the data-parallel
 * implementation needs half the synchronisation points). */
void orc_tick(orc_env *e) {
    tls_events(e);
    build_lists(e);
    register_approaches(e);
    plan(e);                /* next speeds */
    lane_change(e);         /* lane-change decisions + cooperation requests ... */
    insertion_check(e);     /* ... and which departure lanes have room, all on the same state */
    move(e);                /* sideways (lane change), forward, lane hand-over, arrival */
    insertion_apply(e);
    e->t += 1;
    e->stats[9] += 1;
}

/* ------------------------------------------------------------------ observe + state/reward */
void orc_observe(orc_env *e) {
    const orc_scenario *sc = e->sc;
    int32_t O = sc->n_obs, S = sc->n_signals;
    for (int32_t i = 0; i < O; ++i) { e->agg_q[i] = e->agg_a[i] = e->agg_w[i] = e->agg_m[i] = 0; e->agg_s[i] = 0; e->lane_arr[i] = 0; }
    /* obs-lane -> signal */
    for (int32_t s = 0; s < e->hw; ++s) {
        int32_t k = e->trip[s];
        if (e->lane[s] >= LANE_PENDING) continue;
        int32_t lane = e->lane[s];
        int32_t oi = sc->lane_obs[lane];
        int detect = 0;
        if (oi >= 0) {
            int32_t route = sc->trip_route[k];
            float d = (sc->lane_len[lane] - e->pos[s]) + sc->route_tlsdist[sc->route_start[route] + e->cursor[s]];
            detect = d <= e->p.max_distance;
        }
        if (!detect) {
            if (e->owner[s] != OWNER_NONE) e->sig_dep[e->owner[s]] += 1;
            e->owner[s] = OWNER_NONE; e->resco_wait[s] = 0; continue;
        }
        int32_t sig = 0;
        while (sc->sig_obs_start[sig + 1] <= oi) sig += 1;
        /* RESCO waiting-time rule (traffic_signal.py:198-202, 222-232) */
        if (e->owner[s] != (uint8_t)sig) {
            e->resco_wait[s] = 0;
            e->sig_arr[sig] += 1;
            e->lane_arr[oi] += 1;
            if (e->owner[s] != OWNER_NONE) e->sig_dep[e->owner[s]] += 1;
        }
        if (e->resco_wait[s] > 0) {
            uint32_t w = (uint32_t)e->resco_wait[s] + (uint32_t)sc->step_length;
            e->resco_wait[s] = (uint16_t)(w > 65535u ? 65535u : w);
        } else if (e->sumo_wait[s] > 0) {
            e->resco_wait[s] = e->sumo_wait[s];
        }
        e->owner[s] = (uint8_t)sig;
        int32_t w = e->resco_wait[s];
        if (w > 0) { e->agg_q[oi] += 1; e->agg_w[oi] += w; if (w > e->agg_m[oi]) e->agg_m[oi] = w; }
        else e->agg_a[oi] += 1;
        e->agg_s[oi] += (uint32_t)(e->speed[s] * 65536.0f + 0.5f);      /* Q16 fixed point: order-independent */
    }
    for (int32_t sg = 0; sg < S; ++sg) {
        int32_t ph = e->phase[sg];
        e->out_phase[sg] = ph;
        int32_t o0 = sc->sig_obs_start[sg], o1 = sc->sig_obs_start[sg + 1];
        int32_t tw = 0, tq = 0, mq = 0;
        for (int32_t oi = o0; oi < o1; ++oi) {
            float sp = (float)e->agg_s[oi] * (1.0f / 65536.0f);
            float *la = e->lane_agg + oi * 5, *dn = e->drq_norm + oi * 5;
            la[0] = (float)e->agg_q[oi]; la[1] = (float)e->agg_a[oi]; la[2] = (float)e->agg_w[oi];
            la[3] = (float)e->agg_m[oi]; la[4] = sp;
            /* states.drq_norm (states.py:34-59): one-hot compares LANE POSITION with PHASE INDEX */
            dn[0] = (oi - o0) == ph ? 1.0f : 0.0f;
            dn[1] = (float)e->agg_a[oi] / 28.0f;
            dn[2] = (float)e->agg_w[oi] / 28.0f;
            dn[3] = (float)e->agg_q[oi] / 28.0f;
            dn[4] = sp / 20.0f / 28.0f;
            tw += e->agg_w[oi]; tq += e->agg_q[oi];
            if (e->agg_q[oi] > mq) mq = e->agg_q[oi];
        }
        e->queue_sum[sg] = tq; e->queue_max[sg] = mq;
        e->wait[sg] = -(float)tw;                                         /* rewards.wait (rewards.py:6-14) */
        float wn = -(float)tw / 224.0f;                                   /* rewards.wait_norm (rewards.py:17-25) */
        e->wait_norm[sg] = wn < -4.0f ? -4.0f : (wn > 4.0f ? 4.0f : wn);
        int32_t pr = tq;                                                  /* rewards.pressure (rewards.py:28-41) */
        for (int32_t i = sc->pr_out_start[sg]; i < sc->pr_out_start[sg + 1]; ++i) pr -= e->agg_q[sc->pr_out_idx[i]];
        e->pressure[sg] = -pr;
        e->mplight[sg * 13] = ph;                                         /* states.mplight (states.py:62-80) */
        e->mplight_full[sg * 49] = (float)ph;                              /* states.mplight_full (states.py:83-113) */
        e->out_arr[sg] = e->sig_arr[sg]; e->out_dep[sg] = e->sig_dep[sg]; e->sig_arr[sg] = 0; e->sig_dep[sg] = 0;
        for (int32_t m = 0; m < 12; ++m) {
            int32_t q = 0, wv = 0, twm = 0, apm = 0;
            float last_speed = 0.0f;
            for (int32_t i = sc->mv_in_start[sg * 12 + m]; i < sc->mv_in_start[sg * 12 + m + 1]; ++i) {
                q += e->agg_q[sc->mv_in_idx[i]];
                wv += e->agg_q[sc->mv_in_idx[i]] + e->agg_a[sc->mv_in_idx[i]];
                twm += e->agg_w[sc->mv_in_idx[i]]; apm += e->agg_a[sc->mv_in_idx[i]];
                last_speed = (float)e->agg_s[sc->mv_in_idx[i]] * (1.0f / 65536.0f);      /* states.py:97: total_speed restarts with every lane */
            }
            for (int32_t i = sc->mv_out_start[sg * 12 + m]; i < sc->mv_out_start[sg * 12 + m + 1]; ++i)
                q -= e->agg_q[sc->mv_out_idx[i]];
            e->mplight[sg * 13 + 1 + m] = q;
            e->wave[sg * 12 + m] = wv;                                    /* states.wave (states.py:116-127) */
            { float *mf = e->mplight_full + sg * 49 + 1 + m * 4; mf[0] = (float)q; mf[1] = (float)twm / 28.0f; mf[2] = last_speed; mf[3] = (float)apm / 28.0f; }
        }
    }
}

void orc_step(orc_env *e, const int32_t *actions) {
    const orc_scenario *sc = e->sc;
    int32_t S = sc->n_signals;
    if (!e->p.fixed_program) {
        for (int32_t s = 0; s < S; ++s) {                    /* Signal.prep_phase (traffic_signal.py:176-184) */
            int32_t a = actions[s], cur = e->phase[s], G = sc->tls_ngreen[s];
            if (a < 0 || a >= sc->tls_nphase[s]) { e->next_phase[s] = cur; continue; }
            e->next_phase[s] = a;
            if (cur != a && cur < G && a < G) {
                int32_t y = sc->tls_yellow[sc->tls_yel_off[s] + cur * G + a];
                if (y >= 0) orc_set_phase(e, s, y);
            }
        }
    }
    const int32_t ratio = e->p.step_ratio > 1 ? e->p.step_ratio : 1;      /* step_sim() = step_ratio x simulationStep() (multi_signal.py:102-105) */
    for (int32_t i = 0; i < sc->yellow_length * ratio; ++i) orc_tick(e);
    if (!e->p.fixed_program)
        for (int32_t s = 0; s < S; ++s) orc_set_phase(e, s, e->next_phase[s]);   /* Signal.set_phase (:186-187) */
    for (int32_t i = 0; i < (sc->step_length - sc->yellow_length) * ratio; ++i) orc_tick(e);
    orc_observe(e);
}

/* ------------------------------------------------------------------ getters */
const float *orc_lane_agg(const orc_env *e) { return e->lane_agg; }
const float *orc_drq_norm(const orc_env *e) { return e->drq_norm; }
const int32_t *orc_phase(const orc_env *e) { return e->out_phase; }
const int32_t *orc_mplight(const orc_env *e) { return e->mplight; }
const int32_t *orc_wave(const orc_env *e) { return e->wave; }
const float *orc_wait(const orc_env *e) { return e->wait; }
const float *orc_wait_norm(const orc_env *e) { return e->wait_norm; }
const int32_t *orc_pressure(const orc_env *e) { return e->pressure; }
const int32_t *orc_queue_sum(const orc_env *e) { return e->queue_sum; }
const int32_t *orc_queue_max(const orc_env *e) { return e->queue_max; }
const int32_t *orc_arrivals(const orc_env *e) { return e->out_arr; }
const int32_t *orc_lane_arrivals(const orc_env *e) { return e->lane_arr; }
const int32_t *orc_departures(const orc_env *e) { return e->out_dep; }
const float *orc_mplight_full(const orc_env *e) { return e->mplight_full; }
/* fresh Signal objects on the running simulation (what MultiSignal.reset does, multi_signal.py:141-147): the RESCO
 * waiting-time bookkeeping starts over and the program is re-installed ([SUMO-K] the current phase restarts) */
void orc_reinit_signals(orc_env *e) {
    const orc_scenario *sc = e->sc;
    for (int32_t s = 0; s < sc->capacity; ++s) { e->owner[s] = OWNER_NONE; e->resco_wait[s] = 0; }
    for (int32_t s = 0; s < sc->n_signals; ++s) {
        if (!e->p.fixed_program) e->left[s] = sc->tls_dur[sc->tls_dur_off[s] + e->phase[s]];
        e->next_phase[s] = 0; e->sig_arr[s] = 0; e->sig_dep[s] = 0;
    }
}
void orc_get_vehicles(const orc_env *e, orc_vehicles *o) {
    o->hw = e->hw; o->next_trip = e->n_inserted; o->trip = e->trip; o->lane = e->lane; o->pos = e->pos; o->speed = e->speed;
    o->accel = e->accel; o->time_loss = e->time_loss; o->cursor = e->cursor; o->sumo_wait = e->sumo_wait;
    o->resco_wait = e->resco_wait; o->depart = e->depart; o->owner = e->owner;
}
const int32_t *orc_trip_log(const orc_env *e) { return e->trip_log; }
/* trips that have departed (depart tick < now) but are not on the network yet: count and seconds waited so far
 * (utils/readXML.py:52-68 charges end_time - depart for them); per_lane[n_lanes] (may be NULL) receives the count per departure lane */
void orc_backlog(const orc_env *e, int64_t out[2], int32_t *per_lane) {
    const orc_scenario *sc = e->sc;
    out[0] = 0; out[1] = 0;
    for (int32_t dl = 0; dl < sc->n_lanes; ++dl) {
        int32_t c = 0;
        for (int32_t k = e->dep_next[dl]; k >= 0 && sc->trip_depart[k] < e->t; k = e->trip_next[k]) { c += 1; out[1] += e->t - sc->trip_depart[k]; }
        out[0] += c;
        if (per_lane) per_lane[dl] = c;
    }
}
const uint16_t *orc_wtot(const orc_env *e) { return e->wtot; }
void orc_debug(const orc_env *e, const int32_t **reason, const int32_t **block) { *reason = e->dbg_reason; *block = e->dbg_block; }
void orc_stats(const orc_env *e, int64_t out[11]) {
    memcpy(out, e->stats, sizeof(e->stats));
    /* [7]: trips whose insertion has been tried and failed so far (departed before the last tick, not yet on the network) */
    int32_t hz = e->t - 1 <= e->sc->horizon ? e->t - 1 : e->sc->horizon;
    out[6] = e->n_active; out[7] = (e->t >= 1 ? e->sc->trips_cum[hz] : 0) - e->n_inserted;
}
