//! The reference's planner entry points over `libpp_hip.so`.
//!
//! Names, argument order and return shapes follow ishidur/path_planner: `hybrid_astar_planning`
//! (src/hybrid_astar.rs:89-100), `astar_planning` (src/astar.rs:53-63),
//! `calc_holonomic_heuristic_with_obstacle` (src/astar.rs:134-140), `calc_optimal_path` and
//! `calc_all_paths` (src/reeds_shepp.rs:60-69, 87-96), `is_collision` (src/hybrid_astar.rs:417),
//! with `Config`, `Car`, `HybridAstarPath` and `RSPath` as the reference declares them.  The
//! viewer argument (`rec`) is left out: the library never records.  `Option::None` of the
//! reference is `None` here; where the reference panics (an `unwrap`, an `assert!`, an index out
//! of range) these functions panic too, with the library's message for the status code.
//!
//! The free functions build a context and a device map per call, as the reference rebuilds its
//! KD-tree per call; a service keeps a [`Session`] and plans many queries per launch with
//! [`Session::plan_batch`].
pub mod sys;

use std::ffi::CStr;
use std::ptr;
use sys::*;

/// reeds_shepp.rs:42-58
#[derive(Debug, Clone, Copy)]
pub struct Car {
    pub center2front: f64,
    pub center2back: f64,
    pub width: f64,
    pub wheel_base: f64,
}

impl Car {
    pub fn new(center2front: f64, center2back: f64, width: f64, wheel_base: f64) -> Self {
        Self { center2front, center2back, width, wheel_base }
    }
}

/// hybrid_astar.rs:13-63 (constructor arguments in the reference's order)
#[derive(Debug, Clone, Copy)]
pub struct Config {
    xy_reso: f64,
    yaw_reso: f64,
    pub move_step: f64,
    n_steer: usize,
    collision_check_step: usize,
    extend_bound: f64,
    gear_cost: f64,
    backward_cost: f64,
    steer_change_cost: f64,
    steer_angle_cost: f64,
    h_cost: f64,
    pub car: Car,
    max_steer: f64,
}

impl Config {
    #[allow(clippy::too_many_arguments)]
    pub fn new(
        xy_reso: f64,
        yaw_reso: f64,
        move_step: f64,
        n_steer: usize,
        collision_check_step: usize,
        extend_bound: f64,
        gear_cost: f64,
        backward_cost: f64,
        steer_change_cost: f64,
        steer_angle_cost: f64,
        h_cost: f64,
        car: Car,
        max_steer: f64,
    ) -> Self {
        Config {
            xy_reso,
            yaw_reso,
            move_step,
            n_steer,
            collision_check_step,
            extend_bound,
            gear_cost,
            backward_cost,
            steer_change_cost,
            steer_angle_cost,
            h_cost,
            car,
            max_steer,
        }
    }

    /// The configuration of src/main.rs's hybrid A* demo.
    pub fn demo() -> Self {
        Config::new(
            2.0,
            15.0_f64.to_radians(),
            0.4,
            20,
            5,
            1.4,
            100.0,
            5.0,
            5.0,
            1.0,
            15.0,
            Car::new(4.5, 1.0, 3.0, 3.5),
            0.6,
        )
    }

    fn raw(&self) -> pp_config {
        pp_config {
            xy_reso: self.xy_reso,
            yaw_reso: self.yaw_reso,
            move_step: self.move_step,
            n_steer: self.n_steer as u64,
            collision_check_step: self.collision_check_step as u64,
            extend_bound: self.extend_bound,
            gear_cost: self.gear_cost,
            backward_cost: self.backward_cost,
            steer_change_cost: self.steer_change_cost,
            steer_angle_cost: self.steer_angle_cost,
            h_cost: self.h_cost,
            car: pp_car {
                center2front: self.car.center2front,
                center2back: self.car.center2back,
                width: self.car.width,
                wheel_base: self.car.wheel_base,
            },
            max_steer: self.max_steer,
        }
    }

    /// Largest curvature of the car: tan(max_steer) / wheel_base (hybrid_astar.rs:382).
    pub fn max_curvature(&self) -> f64 {
        self.max_steer.tan() / self.car.wheel_base
    }
}

/// hybrid_astar.rs:80-87
#[derive(Debug)]
pub struct HybridAstarPath {
    pub x: Vec<f64>,
    pub y: Vec<f64>,
    pub yaw: Vec<f64>,
    pub direction: Vec<isize>,
    pub cost: f64,
}

/// reeds_shepp.rs:8-17
#[derive(Clone, Debug)]
pub struct RSPath {
    pub lengths: Vec<f64>,
    pub ctypes: Vec<char>,
    pub total_length: f64,
    pub traj_x: Vec<f64>,
    pub traj_y: Vec<f64>,
    pub traj_yaw: Vec<f64>,
    pub traj_dirs: Vec<isize>,
}

fn message(status: i32) -> String {
    // pp_strerror returns a pointer to a static string for every code
    unsafe { CStr::from_ptr(pp_strerror(status)) }.to_string_lossy().into_owned()
}

/// Panics with the library's message unless `status` is PP_OK (the reference panics in these places).
fn check(status: i32, what: &str) {
    if status != PP_OK {
        panic!("{}: {} ({})", what, message(status), status);
    }
}

/// A device context and an obstacle map kept across calls.
pub struct Session {
    ctx: *mut pp_ctx,
    map: *mut pp_map,
}

// A context is not thread-safe, but it may move between threads.
unsafe impl Send for Session {}

impl Session {
    /// `device`: HIP ordinal.  ox / oy: the obstacle points (hybrid_astar.rs:96-97).
    pub fn new(device: i32, ox: &[f64], oy: &[f64]) -> Self {
        assert_eq!(ox.len(), oy.len());
        let mut ctx = ptr::null_mut();
        check(unsafe { pp_ctx_create(device, ptr::null_mut(), &mut ctx) }, "pp_ctx_create");
        let mut map = ptr::null_mut();
        let st = unsafe { pp_map_create(ctx, ox.as_ptr(), oy.as_ptr(), ox.len() as u64, &mut map) };
        if st != PP_OK {
            unsafe { pp_ctx_destroy(ctx) };
            check(st, "pp_map_create");
        }
        Session { ctx, map }
    }

    /// Per-query cap on closed nodes (the reference has none and does not terminate on an
    /// unreachable goal); a query that reaches it panics with the PP_ERR_LIMIT message.
    pub fn set_max_expansions(&mut self, n: u64) {
        check(unsafe { pp_ctx_set_max_expansions(self.ctx, n) }, "pp_ctx_set_max_expansions");
    }

    /// Worker threads a batch may start beside the calling thread.
    pub fn set_host_threads(&mut self, n: u32) {
        check(unsafe { pp_ctx_set_host_threads(self.ctx, n) }, "pp_ctx_set_host_threads");
    }

    /// hybrid_astar_planning (hybrid_astar.rs:89-216) for one query.
    pub fn plan(&mut self, start: [f64; 3], goal: [f64; 3], conf: &Config) -> Option<HybridAstarPath> {
        self.plan_batch(&[start], &[goal], conf).pop().unwrap()
    }

    /// hybrid_astar_planning for many independent queries advanced together on the device.
    pub fn plan_batch(&mut self, starts: &[[f64; 3]], goals: &[[f64; 3]], conf: &Config) -> Vec<Option<HybridAstarPath>> {
        assert_eq!(starts.len(), goals.len());
        let n = starts.len();
        let c = conf.raw();
        let mut cap = 4096usize;
        loop {
            let mut x = vec![0f64; n * cap];
            let mut y = vec![0f64; n * cap];
            let mut yaw = vec![0f64; n * cap];
            let mut dir = vec![0i64; n * cap];
            let mut out: Vec<pp_path> = (0..n)
                .map(|i| pp_path {
                    x: x[i * cap..].as_mut_ptr(),
                    y: y[i * cap..].as_mut_ptr(),
                    yaw: yaw[i * cap..].as_mut_ptr(),
                    direction: dir[i * cap..].as_mut_ptr(),
                    cap: cap as u64,
                    len: 0,
                    cost: 0.0,
                })
                .collect();
            let mut status = vec![0i32; n];
            let st = unsafe {
                pp_plan_batch(
                    self.ctx,
                    self.map,
                    &c,
                    n as u64,
                    starts.as_ptr() as *const f64,
                    goals.as_ptr() as *const f64,
                    out.as_mut_ptr(),
                    ptr::null_mut(),
                    status.as_mut_ptr(),
                )
            };
            check(st, "pp_plan_batch");
            if let Some(need) = (0..n).filter(|&i| status[i] == PP_ERR_CAPACITY).map(|i| out[i].len as usize).max() {
                cap = need.max(2 * cap);  // a path longer than the buffers: once more with room for it
                continue;
            }
            return (0..n)
                .map(|i| match status[i] {
                    PP_OK => {
                        let (lo, hi) = (i * cap, i * cap + out[i].len as usize);
                        Some(HybridAstarPath {
                            x: x[lo..hi].to_vec(),
                            y: y[lo..hi].to_vec(),
                            yaw: yaw[lo..hi].to_vec(),
                            direction: dir[lo..hi].iter().map(|&d| d as isize).collect(),
                            cost: out[i].cost,
                        })
                    }
                    PP_NO_PATH => None,
                    e => panic!("hybrid_astar_planning, query {}: {} ({})", i, message(e), e),
                })
                .collect();
        }
    }

    /// is_collision (hybrid_astar.rs:417-448): does any pose of the list touch an obstacle.
    pub fn is_collision(&mut self, x: &[f64], y: &[f64], yaw: &[f64], conf: &Config) -> bool {
        assert!(x.len() == y.len() && y.len() == yaw.len());
        let mut hit = 0i32;
        let st = unsafe {
            pp_is_collision(
                self.ctx,
                self.map,
                &conf.raw(),
                x.len() as u64,
                x.as_ptr(),
                y.as_ptr(),
                yaw.as_ptr(),
                &mut hit,
                ptr::null_mut(),
            )
        };
        check(st, "pp_is_collision");
        hit != 0
    }

    /// calc_holonomic_heuristic_with_obstacle (astar.rs:134-198) towards (gx, gy): hmap[x - minx][y - miny].
    pub fn holonomic_heuristic(&mut self, gx: f64, gy: f64, reso: f64, rr: f64) -> Vec<Vec<f64>> {
        let mut dims = [0u64; 6];
        // the size first (a call without room reports the dimensions)
        let st = unsafe { pp_heuristic(self.ctx, self.map, gx, gy, reso, rr, ptr::null_mut(), 0, dims.as_mut_ptr()) };
        if st != PP_ERR_CAPACITY {
            check(st, "pp_heuristic");
        }
        let (xw, yw) = (dims[4] as usize, dims[5] as usize);
        let mut flat = vec![0f64; xw * yw];
        let st = unsafe {
            pp_heuristic(self.ctx, self.map, gx, gy, reso, rr, flat.as_mut_ptr(), flat.len() as u64, dims.as_mut_ptr())
        };
        check(st, "pp_heuristic");
        flat.chunks(yw).map(|col| col.to_vec()).collect()
    }

    /// astar_planning (astar.rs:53-132).
    pub fn astar(&mut self, sx: f64, sy: f64, gx: f64, gy: f64, reso: f64, rr: f64) -> Option<(Vec<f64>, Vec<f64>)> {
        let mut cap = 1usize << 12;
        loop {
            let (mut px, mut py, mut len) = (vec![0f64; cap], vec![0f64; cap], 0u64);
            let st = unsafe {
                pp_astar_plan(
                    self.ctx,
                    self.map,
                    sx,
                    sy,
                    gx,
                    gy,
                    reso,
                    rr,
                    px.as_mut_ptr(),
                    py.as_mut_ptr(),
                    cap as u64,
                    &mut len,
                    ptr::null_mut(),
                    0,
                    ptr::null_mut(),
                )
            };
            match st {
                PP_OK => {
                    px.truncate(len as usize);
                    py.truncate(len as usize);
                    return Some((px, py));
                }
                PP_NO_PATH => return None,
                PP_ERR_CAPACITY => cap = (len as usize).max(2 * cap),
                e => panic!("astar_planning: {} ({})", message(e), e),
            }
        }
    }
}

impl Drop for Session {
    fn drop(&mut self) {
        unsafe {
            pp_map_destroy(self.map);
            pp_ctx_destroy(self.ctx);
        }
    }
}

/// hybrid_astar.rs:89-100, without the viewer argument.
#[allow(clippy::too_many_arguments)]
pub fn hybrid_astar_planning(
    sx: f64,
    sy: f64,
    syaw: f64,
    gx: f64,
    gy: f64,
    gyaw: f64,
    ox: &Vec<f64>,
    oy: &Vec<f64>,
    conf: &Config,
) -> Option<HybridAstarPath> {
    Session::new(0, ox, oy).plan([sx, sy, syaw], [gx, gy, gyaw], conf)
}

/// astar.rs:53-63, without the viewer argument.
#[allow(clippy::too_many_arguments)]
pub fn astar_planning(
    sx: f64,
    sy: f64,
    gx: f64,
    gy: f64,
    ox: &[f64],
    oy: &[f64],
    reso: f64,
    rr: f64,
) -> Option<(Vec<f64>, Vec<f64>)> {
    Session::new(0, ox, oy).astar(sx, sy, gx, gy, reso, rr)
}

/// astar.rs:134-140; the goal is given as its position instead of a `Node` (only the last point
/// of the node's lists is read there, :142-143).
pub fn calc_holonomic_heuristic_with_obstacle(
    gx: f64,
    gy: f64,
    ox: &Vec<f64>,
    oy: &Vec<f64>,
    reso: f64,
    rr: f64,
) -> Vec<Vec<f64>> {
    Session::new(0, ox, oy).holonomic_heuristic(gx, gy, reso, rr)
}

fn rs_path(w: &pp_rs_word, tx: &[f64], ty: &[f64], tyaw: &[f64], tdir: &[i64]) -> RSPath {
    let n = w.nseg as usize;
    RSPath {
        lengths: w.lengths[..n].to_vec(),
        ctypes: w.ctypes[..n].iter().map(|&c| c as char).collect(),
        total_length: w.total_length,
        traj_x: tx.to_vec(),
        traj_y: ty.to_vec(),
        traj_yaw: tyaw.to_vec(),
        traj_dirs: tdir.iter().map(|&d| d as isize).collect(),
    }
}

/// A context without a map, for the Reeds-Shepp functions (they read no obstacles).
struct BareContext(*mut pp_ctx);

impl BareContext {
    fn new() -> Self {
        let mut ctx = ptr::null_mut();
        check(unsafe { pp_ctx_create(0, ptr::null_mut(), &mut ctx) }, "pp_ctx_create");
        BareContext(ctx)
    }
}

impl Drop for BareContext {
    fn drop(&mut self) {
        unsafe { pp_ctx_destroy(self.0) };
    }
}

/// reeds_shepp.rs:87-128
#[allow(clippy::too_many_arguments)]
pub fn calc_all_paths(
    sx: f64,
    sy: f64,
    syaw: f64,
    gx: f64,
    gy: f64,
    gyaw: f64,
    max_curvature: f64,
    step_size: f64,
) -> Vec<RSPath> {
    let ctx = BareContext::new();
    let mut cap = 1usize << 16;
    loop {
        let mut n_paths = 0i32;
        let mut words = vec![pp_rs_word::default(); PP_RS_MAX_PATHS];
        let mut off = vec![0u64; PP_RS_MAX_PATHS + 1];
        let (mut tx, mut ty, mut tyaw, mut tdir) = (vec![0f64; cap], vec![0f64; cap], vec![0f64; cap], vec![0i64; cap]);
        let st = unsafe {
            pp_rs_all_paths(
                ctx.0,
                sx,
                sy,
                syaw,
                gx,
                gy,
                gyaw,
                max_curvature,
                step_size,
                &mut n_paths,
                words.as_mut_ptr(),
                off.as_mut_ptr(),
                tx.as_mut_ptr(),
                ty.as_mut_ptr(),
                tyaw.as_mut_ptr(),
                tdir.as_mut_ptr(),
                cap as u64,
            )
        };
        if st == PP_ERR_CAPACITY {
            cap *= 4;
            continue;
        }
        check(st, "calc_all_paths");
        return (0..n_paths as usize)
            .map(|k| {
                let (lo, hi) = (off[k] as usize, off[k + 1] as usize);
                rs_path(&words[k], &tx[lo..hi], &ty[lo..hi], &tyaw[lo..hi], &tdir[lo..hi])
            })
            .collect();
    }
}

/// reeds_shepp.rs:60-85
#[allow(clippy::too_many_arguments)]
pub fn calc_optimal_path(
    sx: f64,
    sy: f64,
    syaw: f64,
    gx: f64,
    gy: f64,
    gyaw: f64,
    maxc: f64,
    step_size: f64,
) -> RSPath {
    let ctx = BareContext::new();
    let mut cap = 1usize << 14;
    loop {
        let mut word = pp_rs_word::default();
        let mut len = 0u64;
        let (mut tx, mut ty, mut tyaw, mut tdir) = (vec![0f64; cap], vec![0f64; cap], vec![0f64; cap], vec![0i64; cap]);
        let st = unsafe {
            pp_rs_optimal_path(
                ctx.0,
                sx,
                sy,
                syaw,
                gx,
                gy,
                gyaw,
                maxc,
                step_size,
                &mut word,
                tx.as_mut_ptr(),
                ty.as_mut_ptr(),
                tyaw.as_mut_ptr(),
                tdir.as_mut_ptr(),
                cap as u64,
                &mut len,
            )
        };
        if st == PP_ERR_CAPACITY {
            cap = (len as usize).max(4 * cap);
            continue;
        }
        check(st, "calc_optimal_path");
        let n = len as usize;
        return rs_path(&word, &tx[..n], &ty[..n], &tyaw[..n], &tdir[..n]);
    }
}

/// calc_rs_path_cost (hybrid_astar.rs:450-489); host arithmetic inside the library.
pub fn calc_rs_path_cost(rspath: &RSPath, conf: &Config) -> f64 {
    let mut w = pp_rs_word::default();
    let n = rspath.lengths.len().min(PP_RS_MAX_SEG);
    w.lengths[..n].copy_from_slice(&rspath.lengths[..n]);
    for i in 0..n {
        w.ctypes[i] = rspath.ctypes[i] as u8;
    }
    w.nseg = n as u8;
    w.total_length = rspath.total_length;
    unsafe { pp_rs_path_cost(&w, &conf.raw()) }
}
