// src/main.rs's hybrid A* demo (needs an MI355X and libpp_hip.so): the cost the reference's
// Python twin and the CPU oracle give for it.
use path_planner_hip::*;

fn design_obstacles(x: usize, y: usize) -> (Vec<f64>, Vec<f64>) {
    let (mut ox, mut oy) = (Vec::new(), Vec::new());
    for i in 0..x { ox.push(i as f64); oy.push(0.0); }
    for i in 0..x { ox.push(i as f64); oy.push((y - 1) as f64); }
    for i in 0..y { ox.push(0.0); oy.push(i as f64); }
    for i in 0..y { ox.push((x - 1) as f64); oy.push(i as f64); }
    for i in 10..21 { ox.push(i as f64); oy.push(15.0); }
    for i in 0..15 { ox.push(20.0); oy.push(i as f64); }
    for i in 15..30 { ox.push(30.0); oy.push(i as f64); }
    for i in 0..16 { ox.push(40.0); oy.push(i as f64); }
    (ox, oy)
}

#[test]
fn demo_query() {
    let (ox, oy) = design_obstacles(51, 31);
    let conf = Config::demo();
    let path = hybrid_astar_planning(10.0, 7.0, 120.0_f64.to_radians(), 45.0, 20.0, 90.0_f64.to_radians(), &ox, &oy, &conf)
        .expect("the demo query has a path");
    assert!((path.cost - 561.447914710932).abs() < 1e-9 * 561.447914710932);
    assert_eq!(path.x.len(), path.direction.len());
}
