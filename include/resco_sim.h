/*
 * resco_sim.h -- C ABI of the MI355X-native batched traffic-signal simulator.
 *
 * Drop-in boundary: this library replaces the `self.sumo` connection object behind RESCO's
 * MultiSignal (the TraCI subset listed below) for N lock-step environment instances on one GPU.
 * One handle = one GPU = one HIP stream; handles are independent (8 GPUs = 8 handles in 8
 * processes or threads); there is no global state.  Functions return 0 on success or a negative
 * RS_E* code and never throw; rs_last_error() gives the message.  The library owns all device
 * buffers; rs_get_buffer() lends device pointers that stay valid until rs_destroy().
 *
 * Reference interface each entry point replaces (paths relative to the RESCO repository):
 *   rs_create    traci.start(sumo_cmd) + the probe run + Signal() program install
 *                resco_benchmark/multi_signal.py:38-47,72-73,132-137; traffic_signal.py:93-100
 *   rs_reset     MultiSignal.reset(): new simulation, fresh Signal objects, first observe
 *                resco_benchmark/multi_signal.py:107-162
 *   rs_step      MultiSignal.step(): prep_phase all -> yellow ticks -> set_phase all -> green
 *                ticks -> observe all -> state_fn / reward_fn -> calc_metrics
 *                resco_benchmark/multi_signal.py:164-197 (step_sim :102-105 = sumo.simulationStep()),
 *                traffic_signal.py:172-187 (FSM), :189-247 (observe / get_vehicles),
 *                states.py:34-127 (drq_norm, mplight, wave), rewards.py:6-41 (wait, wait_norm, pressure)
 *   rs_get_buffer  the TraCI getters used per step: trafficlight.getPhase (traffic_signal.py:174),
 *                lane.getLastStepVehicleIDs (:240), vehicle.getNextTLS / getWaitingTime / getSpeed /
 *                getAcceleration / getLanePosition / getTypeID (:201-210,241), simulation.getTime
 *                (multi_signal.py:190,212)
 *   rs_act_*     the static agents' act(): agents/maxwave.py:18-38, maxpressure.py:13-18,
 *                stochastic.py:17-18 (batched, on device)
 *   rs_destroy   traci.close()  multi_signal.py:231-234
 */
#ifndef RESCO_SIM_H
#define RESCO_SIM_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RS_OK 0
#define RS_EINVAL (-1)     /* bad argument */
#define RS_EHIP (-2)       /* HIP runtime error (see rs_last_error) */
#define RS_ENOMEM (-3)
#define RS_ELIMIT (-4)     /* scenario exceeds a compiled-in limit (LDS budget, id widths) */

/* Flat scenario tables built by resco_amd/scenario.py (host pointers; copied to the device by
 * rs_create).  Field order is shared with the ctypes mirror in resco_amd/_abi.py. */
typedef struct rs_scenario {
    int32_t n_lanes, n_links, n_edges, n_routes, n_trips, n_signals, n_obs, n_vtypes;
    int32_t n_foes, n_route_steps, n_tls_states, n_tls_dur, n_tls_yellow;
    int32_t n_fix_states, n_fix_dur, n_mv_in, n_mv_out, n_pr_out;
    int32_t horizon, capacity, step_length, yellow_length, kmax;     /* kmax: most lanes of one edge */
    /* capacity = vehicle slots per environment: a multiple of 64 in [64, 1984] (grid cells carry 11-bit slot ids); every
     * vehicle type must satisfy floor(32 m / (length + minGap)) + 1 < 15 (4-bit vehicle counter per 32 m grid cell) --
     * rs_create answers RS_ELIMIT otherwise */
    /* lanes (normal + junction-internal), compact ids */
    const float *lane_len, *lane_vmax;
    const int32_t *lane_edge, *lane_left, *lane_right, *lane_link_start, *lane_link_cnt, *lane_obs, *lane_internal;
    /* links (lane -> next lane) */
    const int32_t *link_to_lane, *link_dest_lane, *link_to_edge, *link_tls, *link_tls_pos, *link_minor, *link_cont;
    const int32_t *link_foe_start, *link_foe_cnt;
    const float *link_via_len;
    const int32_t *link_via1, *link_via2, *link_from_lane, *foe_link;
    /* edges */
    const int32_t *edge_lane0, *edge_nlanes;
    /* routes: CSR over route steps */
    const int32_t *route_start, *route_edge;
    const float *route_tlsdist;
    const float *route_cont;            /* [n_route_steps][kmax]: metres drivable along the route without a lane change (bestLanes) */
    /* demand (identical in every environment) */
    const int32_t *trip_depart, *trip_route, *trip_vtype, *trips_cum;
    const float *vtype_params;          /* [n_vtypes][10]: length minGap accel decel tau sigma maxSpeed sfMean sfDev emergencyDecel */
    /* controlled signals: program installed by Signal.__init__ (greens + generated yellows) */
    const int32_t *tls_nphase, *tls_ngreen, *tls_nlinks, *tls_state_off, *tls_dur_off, *tls_yel_off, *tls_init_phase;
    const int32_t *tls_states, *tls_dur, *tls_yellow;
    /* the net file's own fixed-time programs (FIXED baseline) */
    const int32_t *fix_nphase, *fix_state_off, *fix_dur_off, *fix_init_phase, *fix_init_left, *fix_states, *fix_dur;
    /* observation gather tables (Signal.lanes / lane_sets / lane_sets_outbound / outbound_lanes) */
    const int32_t *obs_lane, *sig_obs_start, *mv_in_start, *mv_in_idx, *mv_out_start, *mv_out_idx;
    const int32_t *pr_out_start, *pr_out_idx;
} rs_scenario;

typedef struct rs_params {
    uint32_t seed;
    float max_distance;       /* detector range, MultiSignal(max_distance=...) */
    float sigma;              /* <0: the vType's Krauss sigma; 0: deterministic parity mode */
    int32_t speed_dev;        /* 1: per-vehicle speedFactor ~ clip(N(mean, dev), 0.2, 2) */
    int32_t fixed_program;    /* 1: run the net's own tlLogic and ignore actions */
    int32_t trip_log;         /* 1: keep a per-trip record (RS_BUF_TRIP_LOG) for tripinfo output; costs N x n_trips x 16 B */
    int32_t step_ratio;       /* simulation steps per step_sim() call (MultiSignal(step_ratio=...), multi_signal.py:102-105): an env-step runs
                               * yellow_length x step_ratio ticks before Signal.set_phase and step_length x step_ratio ticks in all; the
                               * RESCO waiting rule still adds step_length per observe (traffic_signal.py:196).  0 or 1: one.
                               * A tick is always ONE second: a sub-second SUMO step length (what the reference uses step_ratio for)
                               * is not modelled -- k means k one-second ticks per step_sim() */
    int32_t tls_hold;         /* what trafficlight.setPhase (Signal.prep_phase / set_phase, traffic_signal.py:176-187) leaves behind.
                               * 0 (default): SUMO's setPhase -- the phase runs for its PROGRAMME DURATION and the programme then
                               * continues with the next index, i -> i + 1 (mod P) (MSSimpleTrafficLightLogic::changeStepAndDuration
                               * re-schedules the switch `duration` seconds ahead [SUMO-K]); the reference never resets a duration, so
                               * a 6 s green chosen for a 10 s step hands its 7th second to the next phase of the list.
                               * 1: the phase STAYS until the next setPhase -- round 5's calibration variant: it reproduces the
                               * reference-held random-policy figures of five maps better (36 of 42 result cells inside +-35 % against
                               * 29, profiles/r06_reference_bands_both_modes.txt), trained agents land on the same delays either way
                               * (profiles/r06_heldout_both_modes.txt).  It was the default of round 5 (then: tls_expiry = 0); round 6
                               * returns to the documented semantics.  Neither is pinned against a SUMO binary
                               * (tools/sumo_runner.py diff decides it on a box that has SUMO).  The net's own programme
                               * (fixed_program) and the phase installed at reset always run on their durations */
} rs_params;

typedef struct rs_sim *rs_handle;

/* env_base: global index of this handle's first environment (keys the counter-based RNG so that a batch
 * sharded over several GPUs reproduces the single-GPU batch).
 * block_threads selects the workgroup shape and the register budget of the step kernel:
 *    0            default = rs_default_block(capacity, n_envs, device_id): one thread per TWO vehicle slots (capacity / 2, a
 *                 multiple of 64, at most 512) with the 80-VGPR build -- three 512-thread workgroups per CU for a 1024-slot
 *                 scenario --, and MORE waves per environment (up to capacity / 64 + 1) while the device's resident-wave budget
 *                 still holds all n_envs environments at once (small batches: BASELINE configs 2 and 4);
 *    n > 0        n threads (a multiple of 64, <= 1024) with the 64-VGPR build;
 *    -n, n <= 512 n threads with the 128-VGPR build;
 *    -(10000 + n) n threads (<= 768) with the 80-VGPR build;
 *    -(20000 + n) n threads, the build chosen by the library (what rs_default_block returns): 64 VGPRs where four 512-thread
 *                 workgroups fit a CU (working memory <= 40 KiB per environment), else 80.
 * Results do not depend on the choice (tests/test_gpu_parity.py::test_block_sizes_and_register_budgets_bit_exact). */
int rs_create(const rs_scenario *sc, const rs_params *p, int32_t n_envs, int32_t env_base, int32_t device_id,
              int32_t block_threads, rs_handle *out);
/* The block_threads value (in the encoding above) that rs_create's default stands for when `n_envs_on_device` environments of a
 * `capacity`-slot scenario share the GPU.  A caller that splits the environments of one GPU over several handles ("pipes") passes
 * the TOTAL here and the result to every rs_create, so that the shape fits the device and not the share of one handle.
 * 0: bad argument.  (Replaces nothing in the reference: its only sizing knob is the number of trial processes, main.py:40-44.) */
int32_t rs_default_block(int32_t capacity, int32_t n_envs_on_device, int32_t device_id);
void rs_destroy(rs_handle h);
const char *rs_last_error(rs_handle h);     /* h may be NULL: error of the last failed rs_create on this thread */

/* Start a new episode in every environment and run the first observe.
 * stream (here and below): a hipStream_t to launch on, or NULL for the handle's own stream.  That stream owns a hardware queue
 * (hipExtStreamCreateWithCUMask, all CUs enabled -- see DESIGN.md "pipes") and is therefore a BLOCKING stream in HIP's sense: it
 * synchronises implicitly with the legacy NULL stream, which is PyTorch's default stream.  Learner kernels or synchronous hipMemcpy
 * calls issued on the NULL stream serialise with every handle's launches: run agent / learner work that is meant to overlap on
 * non-default streams (torch.cuda.Stream), or set RESCO_PLAIN_STREAMS=1 to get plain hipStreamNonBlocking streams (multiplexed
 * over HIP's four hardware queues).  A
 * caller that works on the default (null) stream - e.g. PyTorch's default stream, whose handle is 0 - must pass
 * hipStreamLegacy ((hipStream_t)1), not 0, to be ordered with its own kernels.  The synchronous calls (rs_sync,
 * rs_read_buffer, rs_stats, rs_snapshot, rs_restore) wait for the handle's stream and for the stream of the most
 * recent launch, which therefore must still exist. */
int rs_reset(rs_handle h, void *stream);
/* One MultiSignal.step() for every environment.  actions: int32 [n_envs][n_signals] (green-phase index per
 * signal), host pointer, or device pointer when actions_on_device != 0; NULL = use the handle's RS_BUF_ACTIONS
 * buffer as is (e.g. filled by rs_act_*).  Asynchronous: outputs are ready after rs_sync / stream sync. */
int rs_step(rs_handle h, const int32_t *actions, int32_t actions_on_device, void *stream);
/* n x step_sim() = n x sumo.simulationStep() (multi_signal.py:102-105) without touching the signal FSM, followed by an
 * observe: MultiSignal's `warmup` ticks (multi_signal.py:139-140) and single simulation steps. */
int rs_ticks(rs_handle h, int32_t n_ticks, void *stream);
/* n x MultiSignal.step_sim() (multi_signal.py:102-105) and NOTHING else: the simulation advances, the Signal objects are
 * not touched -- no observe, so Signal.waiting_times, the arrival / departure sets and every output buffer stay as the last
 * observe left them (vehicles that leave meanwhile are counted into the next observe's departures). */
int rs_step_sim(rs_handle h, int32_t n_ticks, void *stream);
/* Which per-lane / per-movement output buffers the observes of the FOLLOWING launches write: bit b of buffer_mask = buffer id
 * b (RS_BUF_LANE_AGG, RS_BUF_DRQ_NORM, RS_BUF_DRQ_NORM_F16, RS_BUF_LANE_ARRIVALS, RS_BUF_MPLIGHT, RS_BUF_WAVE,
 * RS_BUF_MPLIGHT_FULL, and RS_BUF_VEH_ACCEL -- the per-vehicle acceleration of the last tick, which only the Signal views' vehicle
 * dicts read); buffers left out keep their contents and cost no HBM traffic.  The per-signal scalars (phase,
 * wait, wait_norm, pressure, queue_sum / max, arrivals, departures) are always written.  Default: all. */
int rs_set_outputs(rs_handle h, uint64_t buffer_mask);
int rs_sync(rs_handle h);
/* Fresh Signal objects on the RUNNING simulation: what MultiSignal.reset() does after (re)starting SUMO
 * (multi_signal.py:141-147 -> Signal.__init__, traffic_signal.py:28-104): the RESCO waiting-time bookkeeping
 * (Signal.waiting_times / last_step_vehicles) starts over, the program is re-installed (the current phase restarts with
 * its full duration, next_phase = 0), and the first observe runs.  Vehicles stay where they are. */
int rs_reinit_signals(rs_handle h, void *stream);

/* Batched on-device static agents writing RS_BUF_ACTIONS. */
int rs_act_random(rs_handle h, uint32_t step_key, void *stream);       /* STOCHASTIC: U{0..G_s-1} */
/* MAXWAVE / MAXPRESSURE: first maximum over the valid phase pairs of obs[pair0]+obs[pair1].
 * phase_pairs int32 [n_pairs][2]; valid int32 [n_signals][n_pairs] = local action of a pair or -1;
 * order int32 [n_signals][n_pairs] = pair indices in the reference's iteration order, -1 terminated
 * (host pointers, copied on first use, may be NULL afterwards).
 * use_pressure: 1 = mplight[1:] (MAXPRESSURE), 0 = wave (MAXWAVE). */
int rs_act_maxwave(rs_handle h, const int32_t *phase_pairs, int32_t n_pairs, const int32_t *valid,
                   const int32_t *order, int32_t use_pressure, void *stream);

enum rs_buffer {
    RS_BUF_LANE_AGG = 0,   /* f32 [N][n_obs][5]  queue, approach, total_wait, max_wait, speed_sum */
    RS_BUF_DRQ_NORM,       /* f32 [N][n_obs][5]  states.drq_norm rows (signal-major, Signal.lanes order) */
    RS_BUF_PHASE,          /* i32 [N][S]         Signal.phase */
    RS_BUF_MPLIGHT,        /* i32 [N][S][13]     states.mplight */
    RS_BUF_WAVE,           /* i32 [N][S][12]     states.wave */
    RS_BUF_WAIT,           /* f32 [N][S]         rewards.wait */
    RS_BUF_WAIT_NORM,      /* f32 [N][S]         rewards.wait_norm */
    RS_BUF_PRESSURE,       /* i32 [N][S]         rewards.pressure */
    RS_BUF_QUEUE_SUM,      /* i32 [N][S]         calc_metrics queue_lengths */
    RS_BUF_QUEUE_MAX,      /* i32 [N][S]         calc_metrics max_queues */
    RS_BUF_ACTIONS,        /* i32 [N][S]         action staging buffer */
    RS_BUF_ENV,            /* i32 [N][4]         ticks since begin, trips inserted, high-water slot, vehicles on the network */
    RS_BUF_TLS,            /* i32 [N][S][4]      phase, time left, next_phase, |Signal.departures| collected since the last observe */
    RS_BUF_VEH_POS,        /* f32 [N][C] */
    RS_BUF_VEH_SPEED,      /* f32 [N][C] */
    RS_BUF_VEH_ACCEL,      /* f32 [N][C] */
    RS_BUF_VEH_TLOSS,      /* f32 [N][C] */
    RS_BUF_VEH_LANE,       /* u16 [N][C]  0xFFFF free */
    RS_BUF_VEH_TRIP,       /* u16 [N][C]  0xFFFF free */
    RS_BUF_VEH_CURSOR,     /* u16 [N][C] */
    RS_BUF_VEH_SWAIT,      /* u16 [N][C]  SUMO waiting time (s) */
    RS_BUF_VEH_RWAIT,      /* u16 [N][C]  RESCO Signal.waiting_times value (s), 0 = not in the dict */
    RS_BUF_VEH_DEPART,     /* u16 [N][C] */
    RS_BUF_VEH_OWNER,      /* u8  [N][C]  index of the signal that observed the vehicle last, 0xFF none */
    RS_BUF_STATS,          /* i64 [N][12] see rs_stats */
    RS_BUF_DRQ_NORM_F16,   /* f16 [N][S][Lmax][5] zero padded states.drq_norm (IDQN rollout layout) */
    RS_BUF_VEH_SF,         /* f32 [N][C]  per-vehicle speedFactor (written at the insertion; the kernel does not read it back) */
    RS_BUF_VEH_WTOT,       /* u16 [N][C]  total halted seconds of the trip so far (maintained only with trip_log) */
    RS_BUF_TRIP_LOG,       /* i32 [N][n_trips][4] depart tick, arrival tick (0: not arrived), timeLoss (1/1024 s), waiting (s);
                              [N][0][4] when trip_log is off */
    RS_BUF_DEP_NEXT,       /* u16 [N][n_dep] next trip of every departure lane's backlog (0xFFFF: none left); departure lanes are
                              the first lanes of the routes' first edges in ascending lane order */
    RS_BUF_VEH_COOP,       /* u32 [N][C]  cooperation request addressed to the vehicle: trip << 16 | slot of the requester (0xFFFFFFFF none) */
    RS_BUF_VEH_COOPLEAD,   /* u32 [N][C]  the target-lane leader a blocked lane changer falls in behind (same encoding) */
    RS_BUF_ARRIVALS,       /* i32 [N][S]  |Signal.arrivals| of the last observe (traffic_signal.py:222-229) */
    RS_BUF_DEPARTURES,     /* i32 [N][S]  |Signal.departures| of the last observe */
    RS_BUF_MPLIGHT_FULL,   /* f32 [N][S][49] states.mplight_full (states.py:83-113) */
    RS_BUF_LANE_ARRIVALS,  /* i32 [N][n_obs] vehicles of the lane that are in their signal's `arrivals` set of the last observe
                            * (the fringe arrivals of rewards.fma2c, rewards.py:94-97) */
    RS_BUF_VEH_COOP_ODD,   /* u32 [N][C]  the mailboxes are double-buffered by tick parity: requests written in odd ticks ... */
    RS_BUF_VEH_COOPLEAD_ODD, /* u32 [N][C] ... (RS_BUF_VEH_COOP / _COOPLEAD hold those of even ticks) */
    RS_BUF_VEH_MAIL,       /* u32 [N][ceil(C/32)]  one bit per slot: "a cooperation request written in the last tick is waiting in this slot's
                            * mailboxes" (round 6).  Inside a launch the bit travels in the vehicle's working-memory record and a plan reads
                            * its mailboxes only when it is set; the bitmap carries it from one launch to the next */
    RS_BUF_COUNT
};
enum rs_dtype { RS_F32 = 0, RS_I32 = 1, RS_U16 = 2, RS_U8 = 3, RS_F16 = 4, RS_I64 = 5, RS_U32 = 6 };

int rs_get_buffer(rs_handle h, int32_t which, void **dev_ptr, int64_t shape[4], int32_t *ndim, int32_t *dtype);
/* convenience: synchronous device->host copy of a whole buffer */
int rs_read_buffer(rs_handle h, int32_t which, void *host_dst, int64_t nbytes);

/* per env: [0] inserted [1] arrived [2] sum duration(s) [3] sum departDelay(s) [4] sum waiting(s)
 * [5] sum timeLoss (1/1024 s) [6] active now [7] backlog: trips whose insertion was tried and has failed so far
 * [8] sum over ticks of active vehicles [9] ticks [10] insertions refused because all `capacity` slots of the environment were
 * taken (the trip stays in its backlog and enters later: non-zero means the run met the limit of the working memory)
 * [11] violations of the step kernel's classification invariants, counted only by the checking build of the library
 * (-DRS_DEVICE_ASSERT, libresco_sim_check.so); always 0 in the production build, which compiles the checks out */
int rs_stats(rs_handle h, int64_t *host_out /* [n_envs][12] */);

/* environment snapshots (device-resident copies of the SoA state) */
int rs_snapshot(rs_handle h, void **snap);
int rs_restore(rs_handle h, const void *snap);
void rs_snapshot_free(rs_handle h, void *snap);

/* kernel timing on the launch stream (hipEvent pairs around every step kernel while enabled) */
int rs_timing(rs_handle h, int32_t enable);
int rs_timing_read(rs_handle h, float *total_ms, int32_t *launches);   /* syncs; resets the accumulators */

/* new RNG seed for subsequent launches (the reference restarts SUMO with --random every episode,
 * multi_signal.py:127).  Call it right before rs_reset(): a vehicle's speedFactor is a function of (seed, environment, trip) that
 * the kernel RE-COMPUTES at every load instead of reading RS_BUF_VEH_SF back (that buffer is write-only: filled at the insertion
 * for whoever reads it), so a new seed without a reset would change the speed factors of the vehicles already on the network.
 * rs_snapshot stores the seed and rs_restore brings it back with the state. */
int rs_set_seed(rs_handle h, uint32_t seed);

/* in-kernel phase timers (development aid): enable, run steps, then read 16 accumulators of wall_clock64 ticks
 * (100 MHz).  Slots 0-2, 4-6, 11-14: one per phase of the step kernel, summed over all workgroups, barrier wait included
 * (0-2 load / FSM / first registrations, 4 P plan + lane-change decision, 5 C insertion check + TLS events, 6 M move,
 * 11-14 observe / outputs / write-back).  Slots 3, 7-10, 15: what one WAVE spends in a role inside a phase, every 16th
 * environment, sum of ticks in the low 40 bits and number of waves above (7 look-ahead list, 8 lane-change list, 9 slots
 * of P; 10 leavers list, 3 slots, 15 whole wave of M).  Reading also resets. */
int rs_phase_profile(rs_handle h, int32_t enable, uint64_t *host_out16);

/* ---- fused IDQN policy forward (BASELINE config 5) ------------------------------------------
 * Replaces the per-signal Q-network evaluation + LinearDecayEpsilonGreedy action selection of the reference's
 * IDQN agent (resco_benchmark/agents/pfrl_dqn.py:17-46, 62-67): Conv2d(1,64,(2,2)) - ReLU - Flatten -
 * Linear(64*H*4,64) - ReLU - Linear(64,64) - ReLU - Linear(64,A) for every signal, on the fp16 observation tensor
 * RS_BUF_DRQ_NORM_F16 [N][S][lmax][5] the step kernel writes.  Weights arrive pre-packed by
 * resco_amd/agents/idqn_fused.py (fc weights as f16 MFMA B-fragments, see resco_amd/csrc/resco_policy.h):
 *   conv_w f32 [S][64][4], conv_b f32 [S][64], w1 f16 [S][64][hp][2][64][4], b1 f32 [S][64], w2 f16 [S][8][2][64][4],
 *   b2 f32 [S][64], w3 f16 [S][8][64][4], b3 f32 [S][32], n_actions i32 [S];  hp = ceil((lmax - 1) / 2), lmax <= 17.
 * rs_idqn_act: obs / actions (int32 [N][S]) / q (float [N][S][8] or NULL) are DEVICE pointers; mode 0: epsilon-greedy with
 * the counter hash over (seed; env_base + env, signal, step_key) -- env_base = the global index of row 0, as in rs_create, so that
 * the pipes of a split batch draw what the single batch draws and what rs_group_step draws; mode 1: the outputs are logits and the action is drawn from
 * softmax(logits) - the IPPO policy head on the same trunk (resco_benchmark/agents/pfrl_ppo.py:49-64, SoftmaxCategoricalHead); launched on `stream` (same convention as rs_step).  dyn: NULL, or a
 * device pointer to {float epsilon; uint32 step_key} that overrides the two scalar arguments - for replaying a captured
 * HIP graph of the whole env-step (policy kernel + rs_step) with values computed by an earlier node of the graph. */
typedef struct rs_policy *rs_policy_handle;
int rs_idqn_create(int32_t device_id, int32_t n_signals, int32_t lmax, const int32_t *n_actions, const float *conv_w,
                   const float *conv_b, const uint16_t *w1, const float *b1, const uint16_t *w2, const float *b2,
                   const uint16_t *w3, const float *b3, rs_policy_handle *out);
int rs_idqn_act(rs_policy_handle p, const void *obs, int32_t n_envs, int32_t env_base, int32_t mode, float epsilon, uint32_t seed,
                uint32_t step_key, const void *dyn, int32_t *actions, float *q, void *stream);
/* Point the policy at caller-owned DEVICE copies of the packed weights (same layouts as rs_idqn_create; any pointer may
 * be NULL = keep the current one).  The buffers are borrowed: they must stay alive and are read by later rs_idqn_act
 * launches in stream order - a learner re-packs its weights on the device after every update without a host copy. */
int rs_idqn_set_device_weights(rs_policy_handle p, const float *conv_w, const float *conv_b, const uint16_t *w1, const float *b1,
                               const uint16_t *w2, const float *b2, const uint16_t *w3, const float *b3);
/* lanes_per_signal[S]: how many lanes every signal's own network observes (2 .. lmax).  The fc1 rows of the padded lanes are
 * zero in the packed weights (and stay zero under training, resco_amd/agents/idqn_learn.py), so the kernel skips them: the
 * work follows the signals' real head sizes (ingolstadt21: 7.8 lanes on average, padded to 17).  Without this call every
 * signal is evaluated at lmax. */
int rs_idqn_set_lanes(rs_policy_handle p, const int32_t *lanes_per_signal);
void rs_idqn_destroy(rs_policy_handle p);

/* ---- a whole group of handles stepped by ONE call --------------------------------------------------------------
 * The environments of one GPU are split over several handles ("pipes") whose kernels overlap on their own
 * streams; driven from Python that costs two calls through ctypes per pipe and step (agent + rs_step, ~37 us each), which is the
 * limit beyond four pipes and at the small batches of BASELINE config 5.  rs_group_step runs the reference's loop body
 *     act = agent.act(obs); obs, rew, done, info = env.step(act)          (resco_benchmark/main.py:104-108)
 * n_steps times for every handle of the group: per step and handle the agent's kernel and the step kernel, on the handle's own
 * stream, asynchronously.  The agent reads the buffers the handle's LAST observe wrote and writes its RS_BUF_ACTIONS:
 *   RS_AGENT_NONE         rs_step(h, NULL, ...): RS_BUF_ACTIONS as it is
 *   RS_AGENT_RANDOM       rs_act_random(h, step_key + k)                   agents/stochastic.py:17-18
 *   RS_AGENT_MAXWAVE / RS_AGENT_MAXPRESSURE   rs_act_maxwave with the tables a first rs_act_maxwave call installed
 *   RS_AGENT_IDQN         rs_idqn_act(policy, h's RS_BUF_DRQ_NORM_F16, mode, epsilon + k * epsilon_step (>= 0), seed, step_key + k);
 *                         the epsilon-greedy draws are keyed by the GLOBAL environment index (env_base + e), so that a batch
 *                         split over pipes or GPUs draws what the single batch draws
 * Returns the first error (codes as everywhere; rs_last_error of the handle it occurred on). */
enum rs_agent { RS_AGENT_NONE = 0, RS_AGENT_RANDOM = 1, RS_AGENT_MAXWAVE = 2, RS_AGENT_MAXPRESSURE = 3, RS_AGENT_IDQN = 4 };
typedef struct rs_group_agent {
    int32_t kind;               /* enum rs_agent */
    uint32_t step_key;          /* RANDOM, IDQN: key of the call's first step */
    rs_policy_handle policy;    /* IDQN */
    int32_t mode;               /* IDQN: 0 epsilon-greedy, 1 softmax sampling (rs_idqn_act) */
    float epsilon, epsilon_step;
    uint32_t seed;
} rs_group_agent;
int rs_group_step(const rs_handle *handles, int32_t n_handles, const rs_group_agent *agent, int32_t n_steps);

/* static facts */
int rs_info(rs_handle h, int32_t *n_envs, int32_t *block_threads, int32_t *lds_bytes, int32_t *max_lanes_per_signal);

#ifdef __cplusplus
}
#endif
#endif
