// pp_rs.h -- device-side records of the Reeds-Shepp kernels, shared with the planner.
#ifndef PP_RS_H_
#define PP_RS_H_

#include <vector>

#include "pp_internal.h"

// One accepted word, lengths in normalised (curvature) units, stored at its generation index.
struct PpRsCand {
  double len[5];
  double total;
  uint32_t word;  // letters packed 4 bits each: 1 L, 2 S, 3 R
  int32_t nseg;
};

struct PpRsPair {
  uint64_t accepted;  // bit i: candidate with generation index i became a path
  uint32_t flags;     // bit1 guard-band decision, bit2 degenerate pair (the :157 assert)
  int32_t n_paths;
};

// Result of one analytic-expansion shot.
struct PpShotOut {
  double cost;       // calc_rs_path_cost of the accepted path
  int32_t path_idx;  // index in calc_all_paths order, -1 = None
  int32_t gen;       // generation index of that path
  int32_t n_paths;
  uint32_t flags;    // bit0 found, bit1 guard band, bit2 degenerate, bit3 NaN cost, bit4 non-finite pose
  uint32_t poses_handed;  // poses of the popped paths' subsampled lists (handed to is_collision)
  uint32_t pad_;
};

struct PpCostParams {
  double backward_cost, gear_cost, steer_angle_cost, steer_change_cost, max_steer;
};

// d_q: six arrays of n doubles back to back (sx, sy, syaw, gx, gy, gyaw).
void pp_rs_words_launch(pp_ctx* ctx, uint64_t n, const double* d_q, double maxc, PpRsCand* d_cands,
                        PpRsPair* d_pairs);
void pp_rs_shot_launch(pp_ctx* ctx, const pp_map* map, const pp_config* conf, uint32_t n,
                       const double* d_q, double maxc, const PpRsCand* d_cands,
                       const PpRsPair* d_pairs, PpShotOut* d_out, pp_rs_word* d_words = nullptr);
// A planner round in one launch: the Reeds-Shepp words and shots of n nodes and the nodes' motion
// primitives side by side.  For rounds of few nodes; large batches use the separate launches,
// whose shapes suit them.
struct PpExpandPlan;
struct PpExpandDev;
struct PpFtabUpd;
void pp_round_launch(pp_ctx* ctx, const pp_map* map, const pp_config* conf, const PpExpandPlan& plan, uint32_t n,
                     const double* d_q, double maxc, PpShotOut* d_out, const int64_t* d_pdir,
                     const double* d_psteer, const double* d_pcost, const double* d_motion, PpExpandDev eo,
                     uint32_t n_upd = 0, const PpFtabUpd* upd = nullptr, unsigned long long* ftab = nullptr,
                     hipEvent_t done = nullptr);
// done: an event to record at the end of the launch (as hipEventRecord behind it would)
int32_t pp_rs_single(pp_ctx* ctx, const double q[6], double maxc, double step_size, int only_rank,
                     int32_t* n_paths, pp_rs_word* words, uint64_t* traj_off, double* traj_x,
                     double* traj_y, double* traj_yaw, int64_t* traj_dir, uint64_t traj_cap,
                     uint32_t* flags_out);
// Trajectory of the path with rank sel[i] of each of n pairs (q6 = six host arrays of n), tightly
// packed: pair i's points are [offs[i], offs[i + 1]) of the output arrays, which live in the
// context's pinned buffer until its next use.
int32_t pp_rs_selected_batch(pp_ctx* ctx, uint32_t n, const double* q6, double maxc, double step_size,
                             const int32_t* sel, std::vector<uint64_t>& offs, std::vector<double>& total,
                             std::vector<uint32_t>& flags, const double** tx, const double** ty,
                             const double** tyaw, const int64_t** tdir);

#endif
