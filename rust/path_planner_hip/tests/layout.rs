// The C structs as the header lays them out (sizes asserted on the Python side as well).
use path_planner_hip::sys::*;
use std::mem::size_of;

#[test]
fn struct_sizes_match_the_header() {
    assert_eq!(size_of::<pp_car>(), 32);
    assert_eq!(size_of::<pp_config>(), 128);
    assert_eq!(size_of::<pp_param>(), 72);
    assert_eq!(size_of::<pp_rs_word>(), 56);
    assert_eq!(size_of::<pp_path>(), 56);
    assert_eq!(size_of::<pp_plan_stats>(), 96);
    assert_eq!(size_of::<pp_expand_out>(), 64);
}
