/*
 * resco_model.h -- constants of the microsimulation model, shared by the HIP kernels (resco_amd/csrc) and by the CPU
 * oracle (oracle/resco_oracle.c, test infrastructure).  Only numbers live here: the two implementations are written
 * independently (a data-parallel one and a sequential one) and must agree bit for bit on every output.
 *
 * Everything tagged [SUMO-K] restates SUMO behaviour from general knowledge of its published model; SUMO itself is not
 * available here, so these values are PARITY-UNPINNED against SUMO and calibrated against the delay figures the
 * reference publishes (resco_benchmark/utils/avg_timeLoss.py), see DESIGN.md section 2.  Every constant can be overridden on the
 * compiler command line (-DRM_...=...): that is how the coordinate search of round 3 varied them; the shipped values are the defaults.
 */
#ifndef RESCO_MODEL_H
#define RESCO_MODEL_H

#ifndef RM_HALT_SPEED
#define RM_HALT_SPEED 0.1f        /* [SUMO-K] a vehicle at or below this speed is "waiting" (getWaitingTime) */
#endif
#ifndef RM_STOP_OFFSET
#define RM_STOP_OFFSET 1.0f       /* metres kept to a stop line */
#endif
#ifndef RM_FOE_GAP_Q
#define RM_FOE_GAP_Q 40           /* a prohibitor arriving within 4.0 s (units of 0.1 s) closes a minor link */
#endif
#ifndef RM_VIS_DIST
#define RM_VIS_DIST 4.5f          /* [SUMO-K] foe visibility distance: a minor link is approached ready to stop until this close */
#endif
#ifndef RM_MAX_HOPS
#define RM_MAX_HOPS 6             /* links examined ahead of a vehicle */
#endif
#ifndef RM_NB_WINDOW
#define RM_NB_WINDOW 128.0f       /* neighbours further away than this (front to front) play no role in a lane change */
#endif
#ifndef RM_SG_ADVANTAGE
#define RM_SG_ADVANTAGE 20.0f     /* speed-gain change: metres of extra room needed on the neighbour lane */
#endif
#ifndef RM_URGENT_DIST
#define RM_URGENT_DIST 80.0f      /* a strategic change this close to the end of the drivable lane accepts tight gaps */
#endif
#ifndef RM_COOP_RANGE
#define RM_COOP_RANGE 80.0f       /* a blocked changer asks the nearest vehicle this far behind it on the target lane to let it in */
#endif
#ifndef RM_LOOK_TIME
#define RM_LOOK_TIME 8.0f         /* [SUMO-K] LC2013 LOOK_FORWARD (10 s there): strategic look-ahead = max(speed, RM_LOOK_MIN_SPEED) * this + RM_LOOK_BASE per lane to cross */
#endif
#ifndef RM_LOOK_BASE
#define RM_LOOK_BASE 10.0f
#endif
#ifndef RM_LOOK_MIN_SPEED
#define RM_LOOK_MIN_SPEED 5.0f
#endif
#ifndef RM_SG_EXTRA_LANES
#define RM_SG_EXTRA_LANES 2       /* [SUMO-K] LC2013: leave the best lanes for speed gain only if (lanes to cross + 2) look-aheads remain */
#endif
#ifndef RM_GOOD_CONT
#define RM_GOOD_CONT 200.0f       /* a connection whose destination lane can be followed this far is as good as the best one */
#endif
#ifndef RM_MIN_LC_LEN
#define RM_MIN_LC_LEN 5.0f        /* an edge shorter than this cannot host a lane change */
#endif
#ifndef RM_CONT_EPS
#define RM_CONT_EPS 0.5f
#endif
#ifndef RM_SWAP_WAIT
#define RM_SWAP_WAIT 20           /* a mutual block (two stationary vehicles side by side, each in the lane the other needs) is
                                     broken up by trading places once both have stood this many seconds ... */
#endif
#ifndef RM_SWAP_EVERY
#define RM_SWAP_EVERY 4           /* ... looked for on every 4th tick only */
#endif
#define RM_BIGF 1.0e30f
#define RM_TLS_HOLD_TICKS 0x3FFFFFFF   /* time left of a phase that does not expire (rs_params.tls_hold = 1): more ticks than any run has */
#ifndef RM_OCC_FACTOR
#define RM_OCC_FACTOR 1.0f        /* [SUMO-K] LC2013 JAM_FACTOR: weight of the target lane's occupation in the usable distance */
#endif
#ifndef RM_SF_QUANT
#define RM_SF_QUANT 4096.0f      /* speed factors are multiples of 1 / 4096 (they fit 16 bits next to the vehicle's position) */
#endif

#endif
