/*
 * pp_oracle.h -- CPU restatement of the reference planner's hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (path_planner_amd/,
 * include/) may include, link or call this; it is loaded by tests/, by
 * __graft_entry__.smoke() and by bench.py's cpu_baseline leg as the checker.
 *
 * Each function restates one reference function in scalar C++ on glibc libm
 * (the libm Rust's f64 methods lower to on Linux), compiled with
 * -ffp-contract=off so no multiply-add is fused.  The reference ships no tests
 * or golden vectors; the restatement is pinned against outputs of the
 * reference's own Python twin generated in the build container
 * (tests/golden/, tests/golden/make_golden.py) for the functions whose
 * semantics coincide, and is "parity unpinned" for the Rust-only semantics
 * (rounding mode of index computations, heap tie-breaking, the kdtree crate's
 * radius test) -- those follow the Rust source text and are cited inline.
 *
 * Shared plain-data types come from include/pp.h.
 */
#ifndef PP_ORACLE_H_
#define PP_ORACLE_H_

#include "../include/pp.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_map orc_map;

/* Obstacle list + bucket grid (stand-in for the KD-tree as a candidate filter). */
orc_map *orc_map_create(const double *ox, const double *oy, uint64_t n);
void orc_map_destroy(orc_map *m);

/* src/hybrid_astar.rs:522-555 */
int32_t orc_map_param(const orc_map *m, const pp_config *conf, pp_param *out);

/* src/hybrid_astar.rs:417-448, one flag per pose.  mode 0: every obstacle is a
 * candidate (the KD-tree query replaced by a full scan); mode 1: bucket-grid
 * candidates.  Both apply dist <= bound_radius, then the rectangle test. */
int32_t orc_collision_flags(const orc_map *m, const pp_config *conf, uint64_t n, const double *x,
                            const double *y, const double *yaw, uint8_t *flags, int32_t mode);
/* Same, OpenMP over poses (for the all-cores CPU baseline). */
int32_t orc_collision_flags_mt(const orc_map *m, const pp_config *conf, uint64_t n,
                               const double *x, const double *y, const double *yaw,
                               uint8_t *flags, int32_t threads);
/* The same with the candidate search chosen: mode 0 brute force, 1 bucket grid, 2 the KD-tree
 * radius query + per-obstacle rot2d of hybrid_astar.rs:426-445 (the reference's own form). */
int32_t orc_collision_flags_mt_mode(const orc_map *m, const pp_config *conf, uint64_t n,
                                    const double *x, const double *y, const double *yaw,
                                    uint8_t *flags, int32_t threads, int32_t mode);

/* src/hybrid_astar.rs:495-514 */
int32_t orc_motion_set(const pp_config *conf, double *steer, double *direc, uint64_t cap,
                       uint64_t *n_out);
/* src/hybrid_astar.rs:252-320 (+ :322-343) for n parents x all primitives */
int32_t orc_expand_batch(const orc_map *m, const pp_config *conf, uint64_t n, const double *px,
                         const double *py, const double *pyaw, const int64_t *pdir,
                         const double *psteer, const double *pcost, pp_expand_out *out);
/* src/hybrid_astar.rs:266-274 */
int32_t orc_rollout_points(const pp_config *conf, double x0, double y0, double yaw0, double u,
                           double d, double *x, double *y, double *yaw, uint64_t cap,
                           uint64_t *n_out);

/* src/astar.rs:244-261.  mode 0: the reference's triple loop; mode 1: scatter
 * per obstacle over a conservative window with the same predicate. */
int32_t orc_obsmap(const double *ox, const double *oy, uint64_t n, double reso, double rr,
                   uint8_t *obsmap, uint64_t cap, uint64_t dims[6], int32_t mode);
/* src/astar.rs:134-198 */
int32_t orc_heuristic(const double *ox, const double *oy, uint64_t n, double gx, double gy,
                      double reso, double rr, double *hmap, uint64_t cap, uint64_t dims[6]);
/* src/astar.rs:53-132 + extract_path :263-295 (rec = None).  closed_order: calc_index keys in
 * the order nodes are closed.  Parity of the tie rule is unpinned: the reference holds no output of
 * this planner, and the Python twin's open list (heapdict) breaks ties differently. */
int32_t orc_astar(const double *ox, const double *oy, uint64_t n, double sx, double sy, double gx,
                  double gy, double reso, double rr, double *path_x, double *path_y, uint64_t cap,
                  uint64_t *len, uint64_t *closed_order, uint64_t closed_cap, uint64_t *expansions);
/* src/astar.rs:153-198 on a given raster (minx=miny=0, maxx=xw, maxy=yw). */
int32_t orc_wavefront(uint64_t xw, uint64_t yw, const uint8_t *obs, uint64_t goal_x,
                      uint64_t goal_y, double *cost);

/* src/reeds_shepp.rs:707-725 (+ de-normalisation :124-125), no trajectories */
int32_t orc_rs_words(double sx, double sy, double syaw, double gx, double gy, double gyaw,
                     double maxc, int32_t *n_paths, pp_rs_word *words);
/* src/reeds_shepp.rs:87-128 */
int32_t orc_rs_all_paths(double sx, double sy, double syaw, double gx, double gy, double gyaw,
                         double maxc, double step_size, int32_t *n_paths, pp_rs_word *words,
                         uint64_t *traj_off, double *traj_x, double *traj_y, double *traj_yaw,
                         int64_t *traj_dir, uint64_t traj_cap);
/* generate_path + the :72-82 selection rule for n pairs */
int32_t orc_rs_batch(uint64_t n, const double *sx, const double *sy, const double *syaw,
                     const double *gx, const double *gy, const double *gyaw, double maxc,
                     int32_t *n_paths, int32_t *best_idx, pp_rs_word *best, uint8_t *flags,
                     int32_t threads);
/* generate_path (src/reeds_shepp.rs:707-725) for n pairs: every word of pair i in generation order at
 * words[i * PP_RS_MAX_PATHS ..], n_paths[i] of them (-1: the :157 assert fires) */
int32_t orc_rs_words_batch(uint64_t n, const double *sx, const double *sy, const double *syaw,
                           const double *gx, const double *gy, const double *gyaw, double maxc,
                           int32_t *n_paths, pp_rs_word *words, int32_t threads);
/* src/hybrid_astar.rs:450-489 */
double orc_rs_path_cost(const pp_rs_word *word, const pp_config *conf);
/* src/hybrid_astar.rs:380-415 for n (start, goal) pairs */
int32_t orc_analytic_batch(const orc_map *m, const pp_config *conf, uint64_t n, const double *sx,
                           const double *sy, const double *syaw, const double *gx,
                           const double *gy, const double *gyaw, uint8_t *found,
                           int32_t *path_idx, int32_t *n_paths, pp_rs_word *word,
                           double *rs_cost);

/* src/hybrid_astar.rs:89-250.  orc_set_max_expansions: cap on closed nodes (0 = none, the
 * reference's behaviour; it does not terminate on unreachable goals); PP_ERR_LIMIT when hit. */
void orc_set_max_expansions(uint64_t n);
int32_t orc_plan(const orc_map *m, const pp_config *conf, const double start[3],
                 const double goal[3], pp_path *out, pp_plan_stats *stats);

#ifdef __cplusplus
}
#endif
#endif
