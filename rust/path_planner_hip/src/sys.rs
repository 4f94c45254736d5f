//! Raw declarations of include/pp.h.  One item per declaration of the header, same order.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_void};

pub const PP_OK: i32 = 0;
pub const PP_NO_PATH: i32 = 1;
pub const PP_ERR_INVALID: i32 = -1;
pub const PP_ERR_HIP: i32 = -2;
pub const PP_ERR_NAN_COST: i32 = -3;
pub const PP_ERR_DEGENERATE_RS: i32 = -4;
pub const PP_ERR_EMPTY_RS: i32 = -5;
pub const PP_ERR_BOUNDS: i32 = -6;
pub const PP_ERR_CAPACITY: i32 = -7;
pub const PP_ERR_NONFINITE: i32 = -8;
pub const PP_ERR_EMPTY_MAP: i32 = -9;
pub const PP_ERR_UNSUPPORTED: i32 = -10;
pub const PP_ERR_LIMIT: i32 = -11;

pub const PP_RS_MAX_SEG: usize = 5;
pub const PP_RS_MAX_PATHS: usize = 46;

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct pp_car {
    pub center2front: f64,
    pub center2back: f64,
    pub width: f64,
    pub wheel_base: f64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct pp_config {
    pub xy_reso: f64,
    pub yaw_reso: f64,
    pub move_step: f64,
    pub n_steer: u64,
    pub collision_check_step: u64,
    pub extend_bound: f64,
    pub gear_cost: f64,
    pub backward_cost: f64,
    pub steer_change_cost: f64,
    pub steer_angle_cost: f64,
    pub h_cost: f64,
    pub car: pp_car,
    pub max_steer: f64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct pp_param {
    pub minx: u64,
    pub miny: u64,
    pub minyaw: i64,
    pub maxx: u64,
    pub maxy: u64,
    pub xw: u64,
    pub yw: u64,
    pub xyreso: f64,
    pub yawreso: f64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct pp_rs_word {
    pub lengths: [f64; PP_RS_MAX_SEG],
    pub total_length: f64,
    pub ctypes: [u8; PP_RS_MAX_SEG],
    pub nseg: u8,
    pub pad_: [u8; 2],
}

#[repr(C)]
pub struct pp_path {
    pub x: *mut f64,
    pub y: *mut f64,
    pub yaw: *mut f64,
    pub direction: *mut i64,
    pub cap: u64,
    pub len: u64,
    pub cost: f64,
}

#[repr(C)]
pub struct pp_plan_stats {
    pub expansions: u64,
    pub stale_pops: u64,
    pub rollouts: u64,
    pub poses_checked: u64,
    pub rs_paths: u64,
    pub ambiguous: u64,
    pub closed_cap: u64,
    pub closed_order: *mut u64,
    pub rs_final_length: f64,
    pub t_heuristic_ms: f64,
    pub t_search_ms: f64,
    pub first_ambiguous: u64,
}

#[repr(C)]
pub struct pp_expand_out {
    pub valid: *mut u8,
    pub xind: *mut u64,
    pub yind: *mut u64,
    pub yawind: *mut i64,
    pub cost: *mut f64,
    pub end_x: *mut f64,
    pub end_y: *mut f64,
    pub end_yaw: *mut f64,
}

#[repr(C)]
pub struct pp_ctx {
    _private: [u8; 0],
}
#[repr(C)]
pub struct pp_map {
    _private: [u8; 0],
}
#[repr(C)]
pub struct pp_slab {
    _private: [u8; 0],
}

#[link(name = "pp_hip")]
extern "C" {
    // lifecycle
    pub fn pp_ctx_create(device: i32, stream: *mut c_void, out: *mut *mut pp_ctx) -> i32;
    pub fn pp_ctx_destroy(ctx: *mut pp_ctx);
    pub fn pp_ctx_stream(ctx: *mut pp_ctx) -> *mut c_void;
    pub fn pp_ctx_synchronize(ctx: *mut pp_ctx) -> i32;
    pub fn pp_strerror(status: i32) -> *const c_char;
    pub fn pp_last_hip_error(ctx: *mut pp_ctx) -> i32;
    pub fn pp_ctx_device_bytes(ctx: *const pp_ctx) -> u64;
    pub fn pp_version() -> *const c_char;
    pub fn pp_ctx_set_max_expansions(ctx: *mut pp_ctx, max_expansions: u64) -> i32;
    pub fn pp_ctx_set_heuristic_limit(ctx: *mut pp_ctx, limit: f64) -> i32;
    pub fn pp_ctx_set_heuristic_workspace(ctx: *mut pp_ctx, bytes: u64) -> i32;
    pub fn pp_ctx_set_host_threads(ctx: *mut pp_ctx, n: u32) -> i32;
    pub fn pp_ctx_set_batch_parts(ctx: *mut pp_ctx, parts: u32) -> i32;
    pub fn pp_ctx_set_table_budget(ctx: *mut pp_ctx, bytes: u64) -> i32;
    pub fn pp_map_create(ctx: *mut pp_ctx, ox: *const f64, oy: *const f64, n: u64, out: *mut *mut pp_map) -> i32;
    pub fn pp_map_create_raster(
        ctx: *mut pp_ctx, occ: *const u8, width: u64, height: u64, x0: i64, y0: i64, out: *mut *mut pp_map,
    ) -> i32;
    pub fn pp_map_destroy(map: *mut pp_map);
    pub fn pp_map_num_obstacles(map: *const pp_map) -> u64;
    pub fn pp_map_is_lattice(map: *const pp_map) -> i32;
    pub fn pp_map_param(map: *const pp_map, conf: *const pp_config, out: *mut pp_param) -> i32;

    // is_collision
    pub fn pp_collision_batch(
        ctx: *mut pp_ctx, map: *const pp_map, conf: *const pp_config, n: u64, x: *const f64, y: *const f64,
        yaw: *const f64, flags: *mut u8,
    ) -> i32;
    pub fn pp_collision_batch_dev(
        ctx: *mut pp_ctx, map: *const pp_map, conf: *const pp_config, n: u64, d_x: *const f64, d_y: *const f64,
        d_yaw: *const f64, d_flags: *mut u8,
    ) -> i32;
    pub fn pp_is_collision(
        ctx: *mut pp_ctx, map: *const pp_map, conf: *const pp_config, n: u64, x: *const f64, y: *const f64,
        yaw: *const f64, hit: *mut i32, ambiguous: *mut u64,
    ) -> i32;

    // calc_next_node
    pub fn pp_motion_set(conf: *const pp_config, steer: *mut f64, direc: *mut f64, cap: u64, n_out: *mut u64) -> i32;
    pub fn pp_expand_batch(
        ctx: *mut pp_ctx, map: *const pp_map, conf: *const pp_config, n: u64, px: *const f64, py: *const f64,
        pyaw: *const f64, pdir: *const i64, psteer: *const f64, pcost: *const f64, out: *mut pp_expand_out,
    ) -> i32;
    pub fn pp_expand_batch_dev(
        ctx: *mut pp_ctx, map: *const pp_map, conf: *const pp_config, n: u64, d_px: *const f64, d_py: *const f64,
        d_pyaw: *const f64, d_pdir: *const i64, d_psteer: *const f64, d_pcost: *const f64,
        d_out: *const pp_expand_out,
    ) -> i32;
    pub fn pp_rollout_points(
        ctx: *mut pp_ctx, conf: *const pp_config, x0: f64, y0: f64, yaw0: f64, u: f64, d: f64, x: *mut f64,
        y: *mut f64, yaw: *mut f64, cap: u64, n_out: *mut u64,
    ) -> i32;

    // the heuristic
    pub fn pp_heuristic(
        ctx: *mut pp_ctx, map: *const pp_map, gx: f64, gy: f64, reso: f64, rr: f64, hmap: *mut f64, cap: u64,
        dims: *mut u64,
    ) -> i32;
    pub fn pp_obsmap(
        ctx: *mut pp_ctx, map: *const pp_map, reso: f64, rr: f64, obsmap: *mut u8, cap: u64, dims: *mut u64,
    ) -> i32;
    pub fn pp_wavefront_dev(
        ctx: *mut pp_ctx, xw: u64, yw: u64, d_obs: *const u8, goal_x: u64, goal_y: u64, d_cost: *mut f64,
        iters: *mut u64,
    ) -> i32;
    pub fn pp_ctx_set_relax_band(ctx: *mut pp_ctx, band: f64) -> i32;
    pub fn pp_wavefront_stats(ctx: *mut pp_ctx, out: *mut u64) -> i32;
    pub fn pp_wavefront_slab_dev(
        ctx: *mut pp_ctx, xw: u64, yw: u64, x_begin: u64, x_end: u64, d_obs_slab: *const u8,
        d_cost_slab: *mut f64, seed_goal: i32, goal_x: u64, goal_y: u64, d_halo_lo_new: *const f64,
        d_halo_hi_new: *const f64, changed_lo: *mut i32, changed_hi: *mut i32, iters: *mut u64,
    ) -> i32;
    pub fn pp_slab_create(
        ctx: *mut pp_ctx, xw: u64, yw: u64, x_begin: u64, x_end: u64, d_obs_slab: *const u8, goal_x: u64,
        goal_y: u64, out: *mut *mut pp_slab,
    ) -> i32;
    pub fn pp_slab_relax(
        slab: *mut pp_slab, d_halo_lo_new: *const f64, d_halo_hi_new: *const f64, limit: f64,
        d_edge_lo: *mut f64, d_edge_hi: *mut f64, pending: *mut f64, iters: *mut u64,
    ) -> i32;
    pub fn pp_slab_reset(slab: *mut pp_slab, d_obs_slab: *const u8) -> i32;
    pub fn pp_slab_read(slab: *mut pp_slab, d_cost_slab: *mut f64) -> i32;
    pub fn pp_slab_gather(slab: *mut pp_slab, n: u64, d_x: *const u64, d_y: *const u64, d_out: *mut f64) -> i32;
    pub fn pp_slab_destroy(slab: *mut pp_slab);
    pub fn pp_slab_relax_all(slabs: *const *mut pp_slab, n: u32, band: f64, rounds: *mut u64) -> i32;
    pub fn pp_wavefront_tiled(
        ctxs: *const *mut pp_ctx, n_ctx: u32, xw: u64, yw: u64, obs: *const u8, goal_x: u64, goal_y: u64,
        cost: *mut f64, band: f64, rounds: *mut u64,
    ) -> i32;

    // Reeds-Shepp
    pub fn pp_rs_batch(
        ctx: *mut pp_ctx, n: u64, sx: *const f64, sy: *const f64, syaw: *const f64, gx: *const f64,
        gy: *const f64, gyaw: *const f64, maxc: f64, n_paths: *mut i32, best_idx: *mut i32,
        best: *mut pp_rs_word, flags: *mut u8,
    ) -> i32;
    pub fn pp_rs_batch_dev(
        ctx: *mut pp_ctx, n: u64, d_sx: *const f64, d_sy: *const f64, d_syaw: *const f64, d_gx: *const f64,
        d_gy: *const f64, d_gyaw: *const f64, maxc: f64, d_n_paths: *mut i32, d_best_idx: *mut i32,
        d_best: *mut pp_rs_word, d_flags: *mut u8,
    ) -> i32;
    pub fn pp_rs_words_batch(
        ctx: *mut pp_ctx, n: u64, sx: *const f64, sy: *const f64, syaw: *const f64, gx: *const f64,
        gy: *const f64, gyaw: *const f64, maxc: f64, n_paths: *mut i32, words: *mut pp_rs_word, flags: *mut u8,
    ) -> i32;
    pub fn pp_rs_all_paths(
        ctx: *mut pp_ctx, sx: f64, sy: f64, syaw: f64, gx: f64, gy: f64, gyaw: f64, maxc: f64, step_size: f64,
        n_paths: *mut i32, words: *mut pp_rs_word, traj_off: *mut u64, traj_x: *mut f64, traj_y: *mut f64,
        traj_yaw: *mut f64, traj_dir: *mut i64, traj_cap: u64,
    ) -> i32;
    pub fn pp_rs_optimal_path(
        ctx: *mut pp_ctx, sx: f64, sy: f64, syaw: f64, gx: f64, gy: f64, gyaw: f64, maxc: f64, step_size: f64,
        word: *mut pp_rs_word, traj_x: *mut f64, traj_y: *mut f64, traj_yaw: *mut f64, traj_dir: *mut i64,
        traj_cap: u64, traj_len: *mut u64,
    ) -> i32;

    // analytic expansion
    pub fn pp_rs_path_cost(word: *const pp_rs_word, conf: *const pp_config) -> f64;
    pub fn pp_analytic_batch(
        ctx: *mut pp_ctx, map: *const pp_map, conf: *const pp_config, n: u64, sx: *const f64, sy: *const f64,
        syaw: *const f64, gx: *const f64, gy: *const f64, gyaw: *const f64, found: *mut u8, path_idx: *mut i32,
        n_paths: *mut i32, word: *mut pp_rs_word, rs_cost: *mut f64,
    ) -> i32;
    pub fn pp_analytic_batch_dev(
        ctx: *mut pp_ctx, map: *const pp_map, conf: *const pp_config, n: u64, d_q6: *const f64,
        d_found: *mut u8, d_path_idx: *mut i32, d_n_paths: *mut i32, d_word: *mut pp_rs_word,
        d_rs_cost: *mut f64,
    ) -> i32;
    pub fn pp_analytic_batch_dev_static(
        ctx: *mut pp_ctx, map: *const pp_map, conf: *const pp_config, n: u64, d_q6: *const f64,
        d_found: *mut u8, d_path_idx: *mut i32, d_n_paths: *mut i32, d_word: *mut pp_rs_word,
        d_rs_cost: *mut f64,
    ) -> i32;
    pub fn pp_analytic_static_max() -> u64;

    // the planner
    pub fn pp_plan(
        ctx: *mut pp_ctx, map: *const pp_map, conf: *const pp_config, start: *const f64, goal: *const f64,
        out: *mut pp_path, stats: *mut pp_plan_stats,
    ) -> i32;
    pub fn pp_plan_batch(
        ctx: *mut pp_ctx, map: *const pp_map, conf: *const pp_config, n: u64, starts: *const f64,
        goals: *const f64, out: *mut pp_path, stats: *mut pp_plan_stats, status: *mut i32,
    ) -> i32;
    pub fn pp_plan_batch_multi(
        ctxs: *const *mut pp_ctx, maps: *const *const pp_map, n_ctx: u32, conf: *const pp_config, n: u64,
        starts: *const f64, goals: *const f64, out: *mut pp_path, stats: *mut pp_plan_stats, status: *mut i32,
    ) -> i32;
    pub fn pp_plan_batch_counters(ctx: *mut pp_ctx, out: *mut u64) -> i32;

    // the plain grid planner
    pub fn pp_astar_plan(
        ctx: *mut pp_ctx, map: *const pp_map, sx: f64, sy: f64, gx: f64, gy: f64, reso: f64, rr: f64,
        path_x: *mut f64, path_y: *mut f64, cap: u64, len: *mut u64, closed_order: *mut u64, closed_cap: u64,
        expansions: *mut u64,
    ) -> i32;
    pub fn pp_astar_plan_batch(
        ctx: *mut pp_ctx, map: *const pp_map, n: u64, starts: *const f64, goals: *const f64, reso: f64, rr: f64,
        path_x: *mut f64, path_y: *mut f64, cap: u64, path_off: *mut u64, status: *mut i32,
        expansions: *mut u64,
    ) -> i32;
}
