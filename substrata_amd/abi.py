"""ctypes mirror of include/sgp.h (the C ABI behind Substrata's PhysicsWorld facade).

Struct layouts, constants and prototypes here must stay in lock-step with include/sgp.h; tests/test_abi.py
checks sizes/offsets against the compiled library (sgp_abi_sizeof) and that every declared symbol is exported.
"""
import ctypes as C
import numpy as np

ABI_VERSION = 1

OK, ERR_INVALID, ERR_NO_DEVICE, ERR_CAPACITY, ERR_HIP, ERR_BAD_ID, ERR_REJECTED, ERR_PEER = 0, -1, -2, -3, -4, -5, -6, -7

MOTION_STATIC, MOTION_KINEMATIC, MOTION_DYNAMIC = 0, 1, 2
LAYER_NON_MOVING, LAYER_MOVING, LAYER_NON_MOVING_NON_COLLIDABLE, LAYER_MOVING_NON_COLLIDABLE = 0, 1, 2, 3
NUM_LAYERS = 4
SHAPE_SPHERE, SHAPE_BOX, SHAPE_CAPSULE, SHAPE_HULL, SHAPE_MESH = 0, 1, 2, 3, 4
INVALID_ID = 0xFFFFFFFF

EVENT_ACTIVATED, EVENT_DEACTIVATED, EVENT_ENTERED_WATER, EVENT_CONTACT_ADDED, EVENT_CONTACT_PERSISTED = 0, 1, 2, 3, 4
NUM_STAGES = 8
STAGE_NAMES = ["apply_forces", "broadphase", "narrowphase", "setup", "solve_velocity", "integrate",
               "solve_position", "finalize"]

f32, i32, u32, u64 = C.c_float, C.c_int32, C.c_uint32, C.c_uint64


class Settings(C.Structure):
    _fields_ = [("num_velocity_steps", i32), ("num_position_steps", i32), ("baumgarte", f32),
                ("penetration_slop", f32), ("speculative_contact_distance", f32),
                ("min_velocity_for_restitution", f32), ("max_penetration_distance", f32),
                ("time_before_sleep", f32), ("point_velocity_sleep_threshold", f32),
                ("contact_point_preserve_lambda_max_dist_sq", f32), ("max_linear_velocity", f32),
                ("max_angular_velocity", f32), ("allow_sleeping", i32), ("warm_start", i32), ("use_body_pair_contact_cache", i32),
                ("body_pair_cache_max_delta_position_sq", f32), ("body_pair_cache_cos_max_delta_rotation_div2", f32)]


class WorldDesc(C.Structure):
    _fields_ = [("max_bodies", u32), ("max_body_pairs", u32), ("max_manifolds", u32), ("device", i32),
                ("gravity", f32 * 3), ("large_body_radius", f32), ("settings", Settings)]


class BodyDesc(C.Structure):
    _fields_ = [("pos", f32 * 3), ("rot", f32 * 4), ("lin_vel", f32 * 3), ("ang_vel", f32 * 3),
                ("shape_type", i32), ("shape", f32 * 4), ("motion_type", i32), ("layer", i32),
                ("mass", f32), ("friction", f32), ("restitution", f32), ("gravity_factor", f32),
                ("linear_damping", f32), ("angular_damping", f32), ("is_sensor", i32),
                ("allow_sleeping", i32), ("activate", i32), ("use_zero_linear_drag", i32), ("userdata", u64)]


class BodyState(C.Structure):
    _fields_ = [("pos", f32 * 3), ("rot", f32 * 4), ("lin_vel", f32 * 3), ("ang_vel", f32 * 3),
                ("active", u32), ("underwater", u32), ("submerged_volume", f32), ("id", u32)]


class BodyPose(C.Structure):
    _fields_ = [("pos", f32 * 3), ("id", u32), ("rot", f32 * 4)]


class PoseVel(C.Structure):
    _fields_ = [("pos", f32 * 3), ("rot", f32 * 4), ("lin_vel", f32 * 3), ("ang_vel", f32 * 3)]


PHYSICS_UPDATE_BYTES = 80


class BodyEvent(C.Structure):
    _fields_ = [("id", u32), ("_pad", u32), ("userdata", u64)]


class ContactEvent(C.Structure):
    _fields_ = [("id1", u32), ("id2", u32), ("userdata1", u64), ("userdata2", u64),
                ("lin_vel1", f32 * 3), ("lin_vel2", f32 * 3), ("base_offset", f32 * 3), ("normal", f32 * 3),
                ("num_points", u32), ("rel_points_on1", (f32 * 3) * 4), ("penetration", f32)]


class Ray(C.Structure):
    _fields_ = [("origin", f32 * 3), ("dir", f32 * 3), ("max_t", f32), ("ignore_id", u32), ("collidable_only", u32)]


class Hit(C.Structure):
    _fields_ = [("id", u32), ("t", f32), ("normal", f32 * 3), ("triangle", u32), ("userdata", u64), ("material", u32), ("bary", f32 * 2), ("sub_shape", u32)]


class StepStats(C.Structure):
    _fields_ = [("num_bodies", u32), ("num_active", u32), ("num_pairs", u32), ("num_manifolds", u32),
                ("num_contact_points", u32), ("num_colours", u32), ("num_colour_rounds", u32),
                ("num_overflow_constraints", u32), ("pairs_dropped", u32), ("manifolds_dropped", u32),
                ("num_activated", u32), ("num_deactivated", u32), ("layer_counts", u32 * NUM_LAYERS),
                ("num_cached_manifolds", u32), ("num_component_constraints", u32), ("num_catch_all_constraints", u32),
                ("num_deferred_vehicles", u32), ("num_wake_pairs", u32), ("tile_solver", u32), ("device_bytes", u64)]


NUM_KERNEL_CLASSES = 32


class StepProfile(C.Structure):
    _fields_ = [("stage_ms", f32 * NUM_STAGES), ("total_ms", f32), ("kernel_ms", f32 * NUM_KERNEL_CLASSES),
                ("kernel_launches", u32 * NUM_KERNEL_CLASSES), ("sweep_bodies", u32), ("num_constraints", u32),
                ("num_contact_points", u32), ("num_colours", u32), ("row_layout", u32)]


class GhostRecord(C.Structure):
    _fields_ = [("pos", f32 * 3), ("rot", f32 * 4), ("lin_vel", f32 * 3), ("ang_vel", f32 * 3),
                ("shape_type", i32), ("shape", f32 * 4), ("mass", f32), ("friction", f32), ("restitution", f32),
                ("motion_type", u32), ("global_id", u64), ("userdata", u64), ("gravity_factor", f32), ("linear_damping", f32),
                ("angular_damping", f32), ("flags", u32), ("_pad", u32 * 2)]


GHOST_FLAG_LAYER_MASK, GHOST_FLAG_SENSOR, GHOST_FLAG_ALLOW_SLEEP, GHOST_FLAG_ZERO_DRAG = 0x3, 1 << 2, 1 << 3, 1 << 4


class TilesStats(C.Structure):
    _fields_ = [("exported", u32), ("sent", u32), ("received", u32), ("ghosts", u32), ("emigrated", u32), ("immigrated", u32),
                ("fast_imports", u32), ("slow_imports", u32), ("route_retries", u32), ("comm_ranks", u32), ("exchanges", u32),
                ("comm_init_ms", f32), ("last_exchange_ms", f32), ("total_exchange_ms", f32), ("rebalances", u32), ("device_creates", u32)]


class BodyCounts(C.Structure):
    _fields_ = [("num_bodies", u32), ("max_bodies", u32), ("num_static", u32), ("num_dynamic", u32), ("num_kinematic", u32),
                ("num_active_dynamic", u32), ("num_active_kinematic", u32), ("num_meshes", u32), ("num_hulls", u32), ("reserved_", u32),
                ("shape_bytes", u64)]


class Migration(C.Structure):
    _fields_ = [("userdata", u64), ("old_id", u32), ("new_id", u32), ("direction", u32), ("peer", u32)]


class ConstraintDump(C.Structure):
    """Debug/test view of one contact constraint of the last step (not part of the reference facade)."""
    _fields_ = [("a", u32), ("b", u32), ("colour", i32), ("np", i32), ("n", f32 * 3), ("lam_n", f32 * 4),
                ("lam_t1", f32 * 4), ("lam_t2", f32 * 4), ("bias", f32 * 4)]


MAX_WHEELS, MAX_GEARS = 4, 8
VEHICLE_CONTROLLER_WHEELED, VEHICLE_CONTROLLER_MOTORCYCLE = 0, 1
VEHICLE_TESTER_SPHERE, VEHICLE_TESTER_CYLINDER = 0, 1


class WheelDesc(C.Structure):
    _fields_ = [("position", f32 * 3), ("suspension_dir", f32 * 3), ("steering_axis", f32 * 3), ("wheel_up", f32 * 3),
                ("wheel_forward", f32 * 3), ("suspension_min_length", f32), ("suspension_max_length", f32),
                ("suspension_preload", f32), ("spring_frequency", f32), ("spring_damping", f32), ("radius", f32),
                ("width", f32), ("inertia", f32), ("angular_damping", f32), ("max_steer_angle", f32),
                ("max_brake_torque", f32), ("max_handbrake_torque", f32), ("longitudinal_friction", (f32 * 2) * 3),
                ("lateral_friction", (f32 * 2) * 3)]


class DifferentialDesc(C.Structure):
    _fields_ = [("left_wheel", i32), ("right_wheel", i32), ("differential_ratio", f32), ("left_right_split", f32),
                ("limited_slip_ratio", f32), ("engine_torque_ratio", f32)]


class AntiRollBarDesc(C.Structure):
    _fields_ = [("left_wheel", i32), ("right_wheel", i32), ("stiffness", f32)]


class VehicleDesc(C.Structure):
    _fields_ = [("body", u32), ("num_wheels", u32), ("wheels", WheelDesc * MAX_WHEELS), ("up", f32 * 3),
                ("forward", f32 * 3), ("cast_radius", f32), ("max_slope_angle", f32), ("engine_max_torque", f32),
                ("engine_min_rpm", f32), ("engine_max_rpm", f32), ("engine_inertia", f32),
                ("engine_angular_damping", f32), ("engine_torque_curve", (f32 * 2) * 3), ("num_gears", u32),
                ("num_reverse_gears", u32), ("gear_ratios", f32 * MAX_GEARS), ("reverse_gear_ratios", f32 * MAX_GEARS),
                ("switch_time", f32), ("clutch_release_time", f32), ("switch_latency", f32), ("shift_up_rpm", f32),
                ("shift_down_rpm", f32), ("clutch_strength", f32), ("num_differentials", u32),
                ("differentials", DifferentialDesc * 2), ("differential_limited_slip_ratio", f32),
                ("num_anti_roll_bars", u32), ("anti_roll_bars", AntiRollBarDesc * 2), ("controller_type", u32),
                ("max_lean_angle", f32), ("lean_spring_constant", f32), ("lean_spring_damping", f32),
                ("lean_spring_integration_coefficient", f32), ("lean_spring_integration_decay", f32),
                ("lean_smoothing_factor", f32), ("lean_steering_limit", u32), ("collision_tester", u32)]


class VehicleInput(C.Structure):
    _fields_ = [("forward", f32), ("right", f32), ("brake", f32), ("hand_brake", f32)]


class WheelState(C.Structure):
    _fields_ = [("suspension_length", f32), ("steer_angle", f32), ("rotation_angle", f32), ("angular_velocity", f32),
                ("has_contact", i32), ("contact_body", u32), ("contact_position", f32 * 3), ("contact_normal", f32 * 3),
                ("contact_longitudinal", f32 * 3), ("contact_lateral", f32 * 3), ("contact_point_velocity", f32 * 3),
                ("suspension_lambda", f32), ("longitudinal_lambda", f32), ("lateral_lambda", f32),
                ("longitudinal_slip", f32), ("lateral_slip", f32)]


class VehicleState(C.Structure):
    _fields_ = [("wheels", WheelState * MAX_WHEELS), ("engine_rpm", f32), ("current_gear", i32),
                ("clutch_friction", f32), ("active", i32)]


class HullInfo(C.Structure):
    _fields_ = [("hull_id", u32), ("num_vertices", u32), ("num_faces", u32), ("num_edges", u32), ("com", f32 * 3), ("rot", f32 * 4),
                ("volume", f32), ("unit_inertia", f32 * 3), ("aabb_min", f32 * 3), ("aabb_max", f32 * 3)]


class MeshInfo(C.Structure):
    _fields_ = [("mesh_id", u32), ("num_vertices", u32), ("num_triangles", u32), ("num_nodes", u32), ("aabb_min", f32 * 3), ("aabb_max", f32 * 3)]


class CapsuleQuery(C.Structure):
    _fields_ = [("pos", f32 * 3), ("rot", f32 * 4), ("radius", f32), ("half_height", f32), ("max_separation", f32),
                ("ignore_id", u32), ("collidable_only", u32), ("movement", f32 * 3), ("active_edges", u32)]


class CompoundChild(C.Structure):
    _fields_ = [("shape_type", i32), ("shape", f32 * 4), ("pos", f32 * 3), ("rot", f32 * 4)]


class QueryContact(C.Structure):
    _fields_ = [("query", u32), ("body", u32), ("point", f32 * 3), ("normal", f32 * 3), ("distance", f32),
                ("point_velocity", f32 * 3), ("motion_type", u32), ("is_sensor", u32), ("inv_mass", f32), ("sub_shape", u32),
                ("userdata", u64)]


ABI_SIZEOF_ORDER = ["sgp_settings", "sgp_world_desc", "sgp_body_desc", "sgp_body_state", "sgp_body_event",
                    "sgp_contact_event", "sgp_ray", "sgp_hit", "sgp_step_stats", "sgp_step_profile", "sgp_ghost_record",
                    "sgp_vehicle_desc", "sgp_vehicle_input", "sgp_vehicle_state", "sgp_hull_info",
                    "sgp_capsule_query", "sgp_query_contact", "sgp_mesh_info"]

STRUCTS = {"sgp_settings": Settings, "sgp_world_desc": WorldDesc, "sgp_body_desc": BodyDesc,
           "sgp_body_state": BodyState, "sgp_body_event": BodyEvent, "sgp_contact_event": ContactEvent,
           "sgp_ray": Ray, "sgp_hit": Hit, "sgp_step_stats": StepStats, "sgp_step_profile": StepProfile,
           "sgp_ghost_record": GhostRecord, "sgp_vehicle_desc": VehicleDesc, "sgp_vehicle_input": VehicleInput,
           "sgp_vehicle_state": VehicleState, "sgp_hull_info": HullInfo, "sgp_capsule_query": CapsuleQuery,
           "sgp_query_contact": QueryContact, "sgp_mesh_info": MeshInfo}

body_desc_dtype = np.dtype(BodyDesc)
body_state_dtype = np.dtype(BodyState)
body_pose_dtype = np.dtype(BodyPose)
ghost_dtype = np.dtype(GhostRecord)
contact_event_dtype = np.dtype(ContactEvent)
body_event_dtype = np.dtype(BodyEvent)
constraint_dump_dtype = np.dtype(ConstraintDump)
ray_dtype = np.dtype(Ray)
hit_dtype = np.dtype(Hit)
pose_vel_dtype = np.dtype(PoseVel)
vehicle_input_dtype = np.dtype(VehicleInput)
vehicle_state_dtype = np.dtype(VehicleState)
capsule_query_dtype = np.dtype(CapsuleQuery)
query_contact_dtype = np.dtype(QueryContact)
compound_child_dtype = np.dtype(CompoundChild)
migration_dtype = np.dtype(Migration)

P = C.POINTER
vp = C.c_void_p

# name (without prefix) -> (restype, argtypes); world handle is void*
PROTOTYPES = {
    "init": (C.c_int, []),
    "abi_version": (C.c_int, []),
    "last_error": (C.c_char_p, []),
    "default_settings": (None, [P(Settings)]),
    "default_world_desc": (None, [P(WorldDesc)]),
    "default_body_desc": (None, [P(BodyDesc)]),
    "world_create": (C.c_int, [P(WorldDesc), P(vp)]),
    "world_destroy": (C.c_int, [vp]),
    "body_add": (C.c_int, [vp, P(BodyDesc), P(u32)]),
    "body_add_batch": (C.c_int, [vp, vp, u32, vp]),
    "body_add_compound": (C.c_int, [vp, P(BodyDesc), vp, u32, P(u32)]),
    "body_compound_size": (C.c_int, [vp, u32, P(u32)]),
    "body_remove": (C.c_int, [vp, u32]),
    "body_activate": (C.c_int, [vp, u32]),
    "body_set_layer": (C.c_int, [vp, u32, i32]),
    "body_get_volume": (C.c_int, [vp, u32, P(f32)]),
    "body_get_userdata": (C.c_int, [vp, u32, P(u64)]),
    "body_set_pose_vel": (C.c_int, [vp, u32, P(f32), P(f32), P(f32), P(f32)]),
    "body_set_pose_shape": (C.c_int, [vp, u32, P(f32), P(f32), P(f32)]),
    "body_set_pose_vel_batch": (C.c_int, [vp, vp, vp, u32]),
    "physics_update_encode": (C.c_int, [u64, P(BodyState), C.c_double, vp]),
    "physics_update_decode": (C.c_int, [vp, P(u64), P(PoseVel), P(C.c_double)]),
    "snapshot_queue_create": (C.c_int, [P(vp)]),
    "snapshot_queue_destroy": (C.c_int, [vp]),
    "snapshot_queue_push_wire": (C.c_int, [vp, vp, C.c_double]),
    "snapshot_queue_push": (C.c_int, [vp, u64, P(PoseVel), C.c_double, C.c_double]),
    "snapshot_queue_ownership": (C.c_int, [vp, u64, C.c_double, C.c_double, C.c_int]),
    "snapshot_queue_poll": (C.c_int, [vp, C.c_double, C.c_double, vp, vp, u32, P(u32)]),
    "snapshot_queue_expire": (C.c_int, [vp, C.c_double, C.c_double, P(u32)]),
    "snapshot_queue_peek": (C.c_int, [vp, u64, P(u32), P(u32), P(C.c_double)]),
    "body_set_pos": (C.c_int, [vp, u32, P(f32)]),
    "body_set_vel": (C.c_int, [vp, u32, P(f32), P(f32)]),
    "body_move_kinematic": (C.c_int, [vp, u32, P(f32), P(f32), f32]),
    "body_add_force": (C.c_int, [vp, u32, P(f32)]),
    "body_add_force_at": (C.c_int, [vp, u32, P(f32), P(f32)]),
    "body_add_torque": (C.c_int, [vp, u32, P(f32)]),
    "body_get_state": (C.c_int, [vp, vp, u32, vp]),
    "world_read_states": (C.c_int, [vp, u32, u32, vp]),
    "world_read_active": (C.c_int, [vp, vp, u32, P(u32)]),
    "world_read_active_view": (C.c_int, [vp, P(vp), P(u32)]),
    "world_read_active_poses_view": (C.c_int, [vp, P(vp), P(u32)]),
    "world_set_water": (C.c_int, [vp, C.c_int, f32]),
    "world_set_contact_events": (C.c_int, [vp, C.c_int]),
    "world_step": (C.c_int, [vp, f32]),
    "world_step_n": (C.c_int, [vp, f32, u32]),
    "world_step_profiled": (C.c_int, [vp, f32, P(StepProfile)]),
    "world_stats": (C.c_int, [vp, P(StepStats)]),
    "world_launch_counts": (C.c_int, [vp, P(u32), P(u32), P(u32)]),
    "kernel_class_name": (C.c_char_p, [C.c_int]),
    "abi_sizeof": (C.c_int, [C.c_int]),
    "world_drain_events": (C.c_int, [vp, C.c_int, vp, u32, P(u32)]),
    "world_event_counts": (C.c_int, [vp, P(u32)]),
    "world_num_bodies": (C.c_int, [vp, P(u32)]),
    "world_body_counts": (C.c_int, [vp, P(BodyCounts)]),
    "raycast": (C.c_int, [vp, vp, u32, vp]),
    "collide_capsules": (C.c_int, [vp, vp, u32, vp, u32, P(u32)]),
    "spherecast": (C.c_int, [vp, vp, vp, u32, vp]),
    "world_export_boundary": (C.c_int, [vp, P(f32), P(f32), f32, vp, u32, P(u32)]),
    "world_import_ghosts": (C.c_int, [vp, vp, u32]),
    "tiles_route": (C.c_int, [vp, u32, u32, vp, u32, f32, vp, u32, vp, vp, u32, P(u32)]),
    "tiles_split": (C.c_int, [vp, u32, vp, vp, vp, P(u32), vp, P(u32)]),
    "tiles_unique_id": (C.c_int, [vp]),
    "tiles_create": (C.c_int, [vp, u32, u32, vp, f32, f32, vp, P(vp)]),
    "tiles_destroy": (C.c_int, [vp]),
    "tiles_exchange": (C.c_int, [vp]),
    "tiles_exchange_group": (C.c_int, [vp, u32]),
    "tiles_get_stats": (C.c_int, [vp, P(TilesStats)]),
    "tiles_rebalance": (C.c_int, [vp, u32, u32, u32, C.c_int]),
    "tiles_rebalance_group": (C.c_int, [vp, u32, u32, u32, u32, C.c_int]),
    "tiles_get_boxes": (C.c_int, [vp, vp]),
    "tiles_drain_migrations": (C.c_int, [vp, vp, u32, P(u32)]),
    "world_device_array": (C.c_int, [vp, C.c_int, P(vp), P(u32)]),
    "world_stream": (C.c_int, [vp, P(vp)]),
    "mesh_create": (C.c_int, [vp, vp, u32, vp, u32, P(MeshInfo)]),
    "mesh_create_with_materials": (C.c_int, [vp, vp, u32, vp, u32, vp, P(MeshInfo)]),
    "mesh_destroy": (C.c_int, [vp, u32]),
    "mesh_edge_flags": (C.c_int, [vp, u32, vp, u32]),
    "hull_destroy": (C.c_int, [vp, u32]),
    "hull_create": (C.c_int, [vp, vp, u32, P(HullInfo)]),
    "hull_create_com": (C.c_int, [vp, vp, u32, P(f32), P(HullInfo)]),
    "default_vehicle_desc": (None, [P(VehicleDesc)]),
    "vehicle_create": (C.c_int, [vp, P(VehicleDesc), P(u32)]),
    "vehicle_destroy": (C.c_int, [vp, u32]),
    "vehicle_set_input": (C.c_int, [vp, u32, P(VehicleInput)]),
    "vehicle_set_inputs": (C.c_int, [vp, u32, u32, vp]),
    "vehicle_get_state": (C.c_int, [vp, u32, P(VehicleState)]),
    "vehicle_get_states": (C.c_int, [vp, u32, u32, vp]),
    "vehicle_reset_drivetrain": (C.c_int, [vp, u32, f32, f32]),
    "vehicle_enable_lean_controller": (C.c_int, [vp, u32, C.c_int]),
    # test/debug helper, not a facade entry point
    "world_dump_constraints": (C.c_int, [vp, vp, u32, P(u32)]),
}


def bind(lib, prefix, names=None):
    """Attach argtypes/restype for every prototype the library exports under `prefix`. Returns the bound names."""
    bound = []
    for name, (res, args) in PROTOTYPES.items():
        if names is not None and name not in names:
            continue
        fn = getattr(lib, prefix + name, None)
        if fn is None:
            continue
        fn.restype = res
        fn.argtypes = args
        bound.append(name)
    return bound
