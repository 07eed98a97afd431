/*
 * sgp.h -- C ABI of the MI355X-native rigid-body stepper that sits behind Substrata's
 *          PhysicsWorld / PhysicsObject facade.
 *
 * Every entry point replaces one piece of the reference's physics facade, which today forwards to
 * JoltPhysics v5.3.0 (an un-vendored dependency).  Citations are relative to /root/reference.
 * There is no FFI seam in the reference: the boundary is the C++ class `PhysicsWorld`
 * (gui_client/PhysicsWorld.h:98-218).  This header is what a thin `PhysicsWorld.cpp` binds instead
 * of `#include <Jolt/...>`; INTEGRATION.md shows that binding.
 *
 * Conventions
 *   - every function returns an int status: SGP_OK (0) or a negative SGP_ERR_* code; nothing throws;
 *   - plain pointers and sizes only; all buffers are caller-owned host memory unless a parameter is
 *     documented as a device pointer;
 *   - one world per handle; calls on one world are externally synchronised (the reference calls the
 *     facade from the main thread only, PhysicsWorld.h:135 / GUIClient.cpp:6365-6515);
 *   - z is up, gravity defaults to (0,0,-9.81)                       (PhysicsWorld.cpp:520);
 *   - quaternions are (x,y,z,w), same memory order as Quatf / JPH::Quat (JoltUtils.h:48-56);
 *   - all arithmetic is fp32 (Jolt is built single precision; dt arrives as double and is narrowed,
 *     PhysicsWorld.cpp:1363).
 *
 * There is NO CPU fallback: every function that needs the device returns SGP_ERR_NO_DEVICE when no
 * gfx950 GPU is usable.
 */
#ifndef SGP_H
#define SGP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SGP_ABI_VERSION 1

/* ---- status codes ---------------------------------------------------------------------------- */
#define SGP_OK                 0
#define SGP_ERR_INVALID       -1   /* bad argument / NULL handle                                   */
#define SGP_ERR_NO_DEVICE     -2   /* no usable HIP device (never falls back to the CPU)            */
#define SGP_ERR_CAPACITY      -3   /* max_bodies exceeded (cf. cMaxBodies, PhysicsWorld.cpp:492)    */
#define SGP_ERR_HIP           -4   /* a HIP runtime call failed; see sgp_last_error()               */
#define SGP_ERR_BAD_ID        -5   /* body id is not live                                           */
#define SGP_ERR_REJECTED      -6   /* addObject's silent rejections (PhysicsWorld.cpp:1178-1189)    */
#define SGP_ERR_PEER          -7   /* sgp_tiles_exchange: ANOTHER rank reported a failure in this exchange; every rank returns (nobody is left waiting) */

/* ---- enums (values are ABI) ------------------------------------------------------------------ */
/* Motion type: JPH::EMotionType as chosen at PhysicsWorld.cpp:1209-1217. */
#define SGP_MOTION_STATIC      0
#define SGP_MOTION_KINEMATIC   1
#define SGP_MOTION_DYNAMIC     2

/* Object layers: namespace Layers, PhysicsWorld.h:67-74; collide matrix PhysicsWorld.cpp:151-189. */
#define SGP_LAYER_NON_MOVING                 0
#define SGP_LAYER_MOVING                     1
#define SGP_LAYER_NON_MOVING_NON_COLLIDABLE  2
#define SGP_LAYER_MOVING_NON_COLLIDABLE      3
#define SGP_NUM_LAYERS                       4

/* Shapes the stepper collides natively.  Sizes are FINAL (scale already applied by the caller):
 *   SPHERE : p[0] = radius                      (unit sphere r=0.5 * scale.x, PhysicsWorld.cpp:1221-1227)
 *   BOX    : p[0..2] = half extents             (unit cube half 0.5 * scale,  PhysicsWorld.cpp:1249-1255;
 *                                                ground quad (w/2,w/2,0.5),   PhysicsWorld.cpp:1123)
 *   CAPSULE: p[0] = radius, p[1] = half height of the cylinder part, axis = local z
 *                                               (player capsule, PlayerPhysics.cpp:31-32,74)            */
#define SGP_SHAPE_SPHERE   0
#define SGP_SHAPE_BOX      1
#define SGP_SHAPE_CAPSULE  2
#define SGP_SHAPE_MESH     4   /* static triangle mesh created with sgp_mesh_create; shape[0] = (float) mesh id; static and kinematic bodies (JPH::MeshShape has no mass properties; a scripted object is a kinematic mesh body moved with sgp_body_move_kinematic).
                                  A mesh body occupies three consecutive body ids (the id returned + two internal aliases that carry
                                  the second and third contact manifold of a body touching the mesh from several sides)             */
#define SGP_SHAPE_HULL     3   /* convex hull created with sgp_hull_create; shape[0] = (float) hull id, body frame = the hull's
                                  centre-of-mass / principal-axes frame (see sgp_hull_info)                                  */

#define SGP_INVALID_ID 0xFFFFFFFFu   /* JPH::BodyID() default = invalid (PhysicsObject.h:106)        */

/* ---- settings: Jolt v5.3.0 PhysicsSettings defaults (Substrata never overrides them) ----------- */
typedef struct sgp_settings {
	int32_t num_velocity_steps;              /* 10   */
	int32_t num_position_steps;              /* 2    */
	float   baumgarte;                       /* 0.2  */
	float   penetration_slop;                /* 0.02 */
	float   speculative_contact_distance;    /* 0.02 */
	float   min_velocity_for_restitution;    /* 1.0  */
	float   max_penetration_distance;        /* 0.2  */
	float   time_before_sleep;               /* 0.5  */
	float   point_velocity_sleep_threshold;  /* 0.03 */
	float   contact_point_preserve_lambda_max_dist_sq; /* 0.01^2 */
	float   max_linear_velocity;             /* 500  */
	float   max_angular_velocity;            /* 0.25*pi*60 */
	int32_t allow_sleeping;                  /* 1    */
	int32_t warm_start;                      /* 1    */
	/* the body-pair contact cache: a pair whose relative pose moved less than this since its manifold was last computed reuses the manifold */
	int32_t use_body_pair_contact_cache;                 /* 1                  */
	float   body_pair_cache_max_delta_position_sq;       /* 0.001^2            */
	float   body_pair_cache_cos_max_delta_rotation_div2; /* cos(2 deg / 2)     */
} sgp_settings;

typedef struct sgp_world_desc {
	uint32_t max_bodies;        /* capacity; reference: 65536 (PhysicsWorld.cpp:492)                 */
	uint32_t max_body_pairs;    /* 0 = 16*max_bodies+1024; reference: 65536 in flight (:501)          */
	uint32_t max_manifolds;     /* 0 = 8*max_bodies+1024;  reference: 10240 constraints (:506)        */
	int32_t  device;            /* HIP device ordinal                                                 */
	float    gravity[3];        /* (0,0,-9.81)                                                        */
	float    large_body_radius; /* bodies with bounding radius above this skip the hashed grid; 0 = 4 m */
	sgp_settings settings;
} sgp_world_desc;

/* One body, as PhysicsWorld::addObject builds it (PhysicsWorld.cpp:1169-1311). */
typedef struct sgp_body_desc {
	float    pos[3];
	float    rot[4];            /* x,y,z,w */
	float    lin_vel[3];
	float    ang_vel[3];
	int32_t  shape_type;        /* SGP_SHAPE_* */
	float    shape[4];          /* see SGP_SHAPE_* */
	int32_t  motion_type;       /* SGP_MOTION_* */
	int32_t  layer;             /* SGP_LAYER_* */
	float    mass;              /* clamped to >= 0.001 (PhysicsWorld.cpp:1238); inertia from shape+mass */
	float    friction;          /* clamped to [0,1] (:1236) */
	float    restitution;       /* clamped to [0,1] (:1237) */
	float    gravity_factor;    /* Jolt default 1 */
	float    linear_damping;    /* Jolt default 0.05 */
	float    angular_damping;   /* Jolt default 0.05 */
	int32_t  is_sensor;         /* mIsSensor (:1235) */
	int32_t  allow_sleeping;    /* Jolt default 1 */
	int32_t  activate;          /* reference adds with EActivation::DontActivate, then activateObject() */
	int32_t  use_zero_linear_drag; /* PhysicsObject::use_zero_linear_drag (buoyancy, PhysicsWorld.cpp:1405) */
	uint64_t userdata;          /* mUserData = (uint64)PhysicsObject* (:1241) */
} sgp_body_desc;

/* Read-back record for one body (what GUIClient.cpp:6581-6690 pulls per active object). */
typedef struct sgp_body_state {
	float    pos[3];
	float    rot[4];
	float    lin_vel[3];
	float    ang_vel[3];
	uint32_t active;            /* 1 = awake */
	uint32_t underwater;        /* PhysicsObject::underwater */
	float    submerged_volume;  /* PhysicsObject::last_submerged_volume */
	uint32_t id;
} sgp_body_state;

/* Event kinds for sgp_world_drain_events. */
#define SGP_EVENT_ACTIVATED         0  /* OnBodyActivated   (PhysicsWorld.cpp:1448-1467)  payload: sgp_body_event    */
#define SGP_EVENT_DEACTIVATED       1  /* OnBodyDeactivated (PhysicsWorld.cpp:1471-1486)  payload: sgp_body_event    */
#define SGP_EVENT_ENTERED_WATER     2  /* physicsObjectEnteredWater (:1416-1420)          payload: sgp_body_event    */
#define SGP_EVENT_CONTACT_ADDED     3  /* OnContactAdded     (:1499-1503)                 payload: sgp_contact_event */
#define SGP_EVENT_CONTACT_PERSISTED 4  /* OnContactPersisted (:1516-1520)                 payload: sgp_contact_event */

typedef struct sgp_body_event {
	uint32_t id;
	uint32_t _pad;
	uint64_t userdata;
} sgp_body_event;

/* What GUIClient::contactAdded/Persisted read from JPH::Body / JPH::ContactManifold
 * (GUIClient.cpp:10588-10632): both linear velocities, mBaseOffset, mRelativeContactPointsOn1. */
typedef struct sgp_contact_event {
	uint32_t id1, id2;              /* id1 < id2 */
	uint64_t userdata1, userdata2;
	float    lin_vel1[3], lin_vel2[3];
	float    base_offset[3];        /* = first contact point on body 1 (world) */
	float    normal[3];             /* from body 1 to body 2 */
	uint32_t num_points;            /* 1..4 */
	float    rel_points_on1[4][3];  /* relative to base_offset */
	float    penetration;           /* max over points, > 0 if overlapping */
} sgp_contact_event;

typedef struct sgp_ray {
	float    origin[3];
	float    dir[3];           /* unit */
	float    max_t;
	uint32_t ignore_id;        /* JPH::BodyID ignore_body_id (PhysicsWorld.h:178) */
	uint32_t collidable_only;  /* traceRayAgainstCollidableObs (PhysicsWorld.cpp:1711-1715) */
} sgp_ray;

typedef struct sgp_hit {
	uint32_t id;               /* SGP_INVALID_ID if no hit */
	float    t;                /* RayTraceResult::hit_t */
	float    normal[3];        /* RayTraceResult::hit_normal_ws */
	uint32_t triangle;         /* mesh hits (ray casts): index of the triangle in the caller's order; SGP_INVALID_ID otherwise            */
	uint64_t userdata;         /* -> RayTraceResult::hit_object */
	uint32_t material;         /* mesh hits: the triangle's user data = material index, MeshShape::GetTriangleUserData
	                              (PhysicsWorld.cpp:1700-1704 -> RayTraceResult::hit_mat_index); 0 otherwise                               */
	float    bary[2];          /* mesh hits: barycentric coordinates (u, v) of the hit, point = (1 - u - v) a + u b + v c; 0 otherwise.
	                              (The reference leaves RayTraceResult::coords at 0, :1693; the facade does the same.)                     */
	uint32_t sub_shape;        /* compound bodies (sgp_body_add_compound): index of the child that was hit, `id` is the compound's id; 0 otherwise */
} sgp_hit;

/* Counters of the last step (getDiagnostics, PhysicsWorld.cpp:1578-1604, plus stage sizes). */
typedef struct sgp_step_stats {
	uint32_t num_bodies;
	uint32_t num_active;
	uint32_t num_pairs;
	uint32_t num_manifolds;
	uint32_t num_contact_points;
	uint32_t num_colours;
	uint32_t num_colour_rounds;
	uint32_t num_overflow_constraints;
	uint32_t pairs_dropped;
	uint32_t manifolds_dropped;
	uint32_t num_activated;      /* activation / deactivation events raised since the end of the previous step */
	uint32_t num_deactivated;
	uint32_t layer_counts[SGP_NUM_LAYERS];
	uint32_t num_cached_manifolds;       /* manifolds taken from the body-pair contact cache instead of a collision test */
	uint32_t num_component_constraints;  /* constraints of the high colours, solved component by component in one launch per pass (device launch plan; 0 on the oracle) */
	uint32_t num_catch_all_constraints;  /* ... of them in components too large for a workgroup (solved serially; the plan then takes fewer colours) */
	uint32_t num_deferred_vehicles;      /* vehicles that shared a movable body (a dynamic body under a wheel, a chassis a wheel stands on) with a vehicle of lower index this step: their rows are solved after the others', in index order */
	uint32_t num_wake_pairs;             /* in-step activation: pairs of the bodies this step woke with what was not awake when it began (part of num_pairs) */
	uint32_t tile_solver;                /* 1: this step's velocity iterations ran as the one resident launch of the tile solver; 2: ... and some body made all tiles neighbours */
	uint64_t device_bytes;
} sgp_step_stats;

/* Per-stage device time of the last profiled step, measured with HIP events on the world's stream. */
#define SGP_STAGE_APPLY_FORCES   0
#define SGP_STAGE_BROADPHASE     1
#define SGP_STAGE_NARROWPHASE    2
#define SGP_STAGE_SETUP          3   /* islands, colouring, constraint setup + cache match */
#define SGP_STAGE_SOLVE_VELOCITY 4
#define SGP_STAGE_INTEGRATE      5   /* the body-array sweep (pose integrate + AABB)  */
#define SGP_STAGE_SOLVE_POSITION 6
#define SGP_STAGE_FINALIZE       7   /* AABB refresh, sleep test, islands sleep, buoyancy, events */
#define SGP_NUM_STAGES           8

#define SGP_NUM_KERNEL_CLASSES 32
typedef struct sgp_step_profile {
	float    stage_ms[SGP_NUM_STAGES];
	float    total_ms;              /* first launch -> last launch of the step, device time */
	float    kernel_ms[SGP_NUM_KERNEL_CLASSES];       /* summed launch durations per kernel class (HIP events) */
	uint32_t kernel_launches[SGP_NUM_KERNEL_CLASSES]; /* launches per class; name: sgp_kernel_class_name() */
	uint32_t sweep_bodies;          /* body slots one launch of the body-array sweep covers */
	uint32_t num_constraints;
	uint32_t num_contact_points;
	uint32_t num_colours;
	uint32_t row_layout;            /* what the velocity iterations of this step read per contact point: 0 full rows (192 B), 1 r x axis only (96 B), 2 none (lever arms + effective masses: 40 B per lane) */
} sgp_step_profile;

typedef struct sgp_world sgp_world;

/* ---- lifecycle ------------------------------------------------------------------------------- */
/* PhysicsWorld::init() (PhysicsWorld.cpp:250-273): once per process. Returns the device count found. */
int  sgp_init(void);
int  sgp_abi_version(void);
const char* sgp_last_error(void);
void sgp_default_settings(sgp_settings* out);
void sgp_default_world_desc(sgp_world_desc* out);
void sgp_default_body_desc(sgp_body_desc* out);
/* PhysicsWorld ctor / dtor (PhysicsWorld.cpp:462-543). */
int  sgp_world_create(const sgp_world_desc* desc, sgp_world** out);
int  sgp_world_destroy(sgp_world* w);

/* ---- bodies ---------------------------------------------------------------------------------- */
/* addObject (PhysicsWorld.cpp:1169-1311).  Returns SGP_ERR_REJECTED for |pos|>1e9 etc. */
int  sgp_body_add(sgp_world* w, const sgp_body_desc* d, uint32_t* id_out);
int  sgp_body_add_batch(sgp_world* w, const sgp_body_desc* d, uint32_t n, uint32_t* ids_out);
/* removeObject (:1315-1339) */
int  sgp_body_remove(sgp_world* w, uint32_t id);
/* activateObject (:1342-1346) */
int  sgp_body_activate(sgp_world* w, uint32_t id);
/* Body::GetShape()->GetVolume() through GetBodyLockInterface().TryGetBody() (BoatPhysics.cpp:40-43) */
int  sgp_body_get_volume(sgp_world* w, uint32_t id, float* volume_out);
/* setObjectLayer (:1349-1353) */
int  sgp_body_set_layer(sgp_world* w, uint32_t id, int32_t layer);
/* Every setter below returns SGP_ERR_INVALID for a non-finite argument and leaves the body untouched (the reference asserts finite
 * inputs, PhysicsWorld.cpp:548-556,625,710). */
/* setNewObToWorldTransform(pos,rot,linvel,angvel) (:607-620); SetPositionRotationAndVelocity. Does not activate. */
int  sgp_body_set_pose_vel(sgp_world* w, uint32_t id, const float pos[3], const float rot[4],
                           const float lin_vel[3], const float ang_vel[3]);
/* setNewObToWorldTransform(pos,rot,scale) (:546-604): zero velocity, new final shape size, activates. */
int  sgp_body_set_pose_shape(sgp_world* w, uint32_t id, const float pos[3], const float rot[4],
                             const float shape[4]);
/* setNewPosition (:623-633): SetPosition, DontActivate. */
int  sgp_body_set_pos(sgp_world* w, uint32_t id, const float pos[3]);
/* setLinearAndAngularVelToZero (:649-657) */
int  sgp_body_set_vel(sgp_world* w, uint32_t id, const float lin_vel[3], const float ang_vel[3]);
/* moveKinematicObject (:707-722): BodyInterface::MoveKinematic; no-op for non-kinematic bodies. */
int  sgp_body_move_kinematic(sgp_world* w, uint32_t id, const float target_pos[3], const float target_rot[4], float dt);
/* BodyInterface::AddForce / AddTorque / AddForce(at point) used by HoverCarPhysics.cpp:113-348, BoatPhysics.cpp:221-267.
 * Accumulate until the next step, then cleared. Activate the body. */
int  sgp_body_add_force(sgp_world* w, uint32_t id, const float force[3]);
int  sgp_body_add_force_at(sgp_world* w, uint32_t id, const float force[3], const float point[3]);
int  sgp_body_add_torque(sgp_world* w, uint32_t id, const float torque[3]);
/* Batched form of setNewObToWorldTransform(pos,rot,linvel,angvel) for network physics snapshots (GUIClient.cpp:7474-7478 inserts
 * one snapshot per object per frame; n objects here cost one upload + one kernel). */
typedef struct sgp_pose_vel { float pos[3]; float rot[4]; float lin_vel[3]; float ang_vel[3]; } sgp_pose_vel;
int  sgp_body_set_pose_vel_batch(sgp_world* w, const uint32_t* ids, const sgp_pose_vel* recs, uint32_t n);

/* ObjectPhysicsTransformUpdate payload (Protocol.h:120, written at GUIClient.cpp:7637-7650): little endian,
 * uid u64 | pos 3 x f64 | rot 4 x f32 (x,y,z,w) | lin vel 3 x f32 | ang vel 3 x f32 | client time f64  = 80 bytes. */
#define SGP_PHYSICS_UPDATE_BYTES 80
int  sgp_physics_update_encode(uint64_t uid, const sgp_body_state* st, double client_time, uint8_t out[SGP_PHYSICS_UPDATE_BYTES]);
int  sgp_physics_update_decode(const uint8_t in[SGP_PHYSICS_UPDATE_BYTES], uint64_t* uid_out, sgp_pose_vel* rec_out, double* client_time_out);

/* ---- network physics snapshots: the de-jitter buffer in front of the step -------------------------------------
 * What the reference keeps per WorldObject (shared/WorldObject.h:540-566: a ring of HISTORY_BUF_SIZE = 4 snapshots, next_snapshot_i,
 * next_insertable_snapshot_i, transmission_time_offset) and does with it:
 *   receive    ClientThread.cpp:736-792   an ObjectPhysicsTransformUpdate from the object's physics owner goes into slot next_snapshot_i % 4
 *   ownership  ClientThread.cpp:957-975   ObjectPhysicsOwnershipTaken: transmission_time_offset = global time now - the sender's time of the
 *                                         change (a renewal only sets it when it is still 0); a change of owner drops the queued snapshots
 *   playback   GUIClient.cpp:7462-7493    once per frame and object: the oldest snapshot not yet inserted is due when
 *                                         global_time >= client_time + transmission_time_offset + padding_delay (0.1 s); it is then fed to
 *                                         setNewObToWorldTransform(pos, rot, lin vel, ang vel) -- here: returned for ONE batched
 *                                         sgp_body_set_pose_vel_batch.  At most one snapshot per object and poll, in arrival order; a ring
 *                                         that overflowed (more than 4 pending) plays back what its slots hold now, exactly as the reference's.
 *   expiry     GUIClient.cpp:7443-7452    an object whose last snapshot arrived more than 1 s ago leaves the active set
 * Host-side state only (no device work); one queue serves any number of objects, keyed by their 64-bit uid. */
typedef struct sgp_snapshot_queue sgp_snapshot_queue;
#define SGP_SNAPSHOT_HISTORY 4
int  sgp_snapshot_queue_create(sgp_snapshot_queue** out);
int  sgp_snapshot_queue_destroy(sgp_snapshot_queue* q);
/* receive: a wire record (sgp_physics_update_decode) or an already decoded one; local_time = the receiver's clock at arrival */
int  sgp_snapshot_queue_push_wire(sgp_snapshot_queue* q, const uint8_t msg[SGP_PHYSICS_UPDATE_BYTES], double local_time);
int  sgp_snapshot_queue_push(sgp_snapshot_queue* q, uint64_t uid, const sgp_pose_vel* rec, double client_time, double local_time);
/* ObjectPhysicsOwnershipTaken for uid */
int  sgp_snapshot_queue_ownership(sgp_snapshot_queue* q, uint64_t uid, double global_time_now, double ownership_change_global_time, int renewal);
/* playback: the snapshots due at global_time, at most one per object, ascending uid; *n_out = how many were due (<= cap are written and consumed) */
int  sgp_snapshot_queue_poll(sgp_snapshot_queue* q, double global_time, double padding_delay,
                             uint64_t* uids_out, sgp_pose_vel* recs_out, uint32_t cap, uint32_t* n_out);
/* expiry: forget the objects whose last snapshot arrived before local_time_now - max_age (reference: 1.0 s); *n_out = objects still tracked */
int  sgp_snapshot_queue_expire(sgp_snapshot_queue* q, double local_time_now, double max_age, uint32_t* n_out);
/* inspection (tests): ring indices and offset of one object; SGP_ERR_BAD_ID when the uid is not tracked */
int  sgp_snapshot_queue_peek(sgp_snapshot_queue* q, uint64_t uid, uint32_t* next_snapshot_i, uint32_t* next_insertable_snapshot_i, double* transmission_time_offset);

/* getObjectLinearVelocity (:636-646), getPosInJolt (:1625-1632), GUIClient.cpp:6588,6673 read-back. */
int  sgp_body_get_state(sgp_world* w, const uint32_t* ids, uint32_t n, sgp_body_state* out);
/* All live bodies in id order [first, first+n). Slots that are not live get id = SGP_INVALID_ID. */
int  sgp_world_read_states(sgp_world* w, uint32_t first, uint32_t n, sgp_body_state* out);
/* The per-frame read-back loop (GUIClient.cpp:6581-6690): compacted states of every ACTIVE body. */
int  sgp_world_read_active(sgp_world* w, sgp_body_state* out, uint32_t cap, uint32_t* n_out);
/* The same without the copy into the caller's buffer: *view_out points at a pinned host buffer of the library holding *n_out records.  The
 * buffer belongs to the two *_view calls alone: it stays valid and unchanged across steps, ray casts, queries and state reads, until the next
 * sgp_world_read_active_view / sgp_world_read_active_poses_view on this world or sgp_world_destroy (the loop at GUIClient.cpp:6581-6690
 * reads each record once, and may trace rays while it does). */
int  sgp_world_read_active_view(sgp_world* w, const sgp_body_state** view_out, uint32_t* n_out);
/* ... and with nothing but what that loop reads per activated body (GetPositionAndRotation, GUIClient.cpp:6586-6588): 32 bytes instead of 68. */
typedef struct sgp_body_pose {
	float    pos[3];
	uint32_t id;
	float    rot[4];            /* x, y, z, w */
} sgp_body_pose;
int  sgp_world_read_active_poses_view(sgp_world* w, const sgp_body_pose** view_out, uint32_t* n_out);

/* ---- world state ----------------------------------------------------------------------------- */
/* setWaterBuoyancyEnabled / setWaterZ (PhysicsWorld.h:109-112) */
int  sgp_world_set_water(sgp_world* w, int enabled, float water_z);
/* Enable capture of contact added/persisted events (only needed when an event_listener is set). */
int  sgp_world_set_contact_events(sgp_world* w, int enabled);
/* think(dt) (PhysicsWorld.cpp:1356-1443): one PhysicsSystem::Update with 1 collision step + buoyancy sweep.
 * Blocks until the step has finished on the device.  A sleeping body that an awake one touches (or a wheel stands on) wakes up IN this
 * step, together with the island it fell asleep with, and collides in it (PhysicsSystem::JobFindCollisions appends what it wakes to the
 * active list); sgp_step_stats::num_wake_pairs counts the pairs that brings.  A step in which nothing is awake and nothing was edited
 * costs nothing (and forgets the previous step's contacts). */
int  sgp_world_step(sgp_world* w, float dt);
/* Same, n steps back to back with a single host sync at the end (GUIClient's sub-step loop, GUIClient.cpp:6382). */
int  sgp_world_step_n(sgp_world* w, float dt, uint32_t n);
/* Same as sgp_world_step but brackets every stage with HIP events on the world's stream. */
int  sgp_world_step_profiled(sgp_world* w, float dt, sgp_step_profile* out);
int  sgp_world_stats(sgp_world* w, sgp_step_stats* out);
/* How the steps so far were issued: replayed as a captured hipGraph, launched eagerly, or skipped because nothing was awake
 * (diagnostic; any pointer may be NULL). */
int  sgp_world_launch_counts(sgp_world* w, uint32_t* graph_replays_out, uint32_t* eager_steps_out, uint32_t* idle_steps_out);
/* Name of kernel class k of sgp_step_profile (NULL past the last class). */
const char* sgp_kernel_class_name(int k);
/* sizeof() of ABI struct number `which` (order: settings, world_desc, body_desc, body_state, body_event, contact_event,
 * ray, hit, step_stats, step_profile, ghost_record, vehicle_desc, vehicle_input, vehicle_state, hull_info, capsule_query,
 * query_contact, mesh_info) so bindings can verify their layout. */
int  sgp_abi_sizeof(int which);
/* activated_obs / newly_activated_obs maintenance + listener callbacks (PhysicsWorld.h:194-200). */
int  sgp_world_drain_events(sgp_world* w, int kind, void* out, uint32_t cap, uint32_t* n_out);
/* Events waiting per kind (index = SGP_EVENT_*), nothing drained: lets the caller size its buffers to what a step produced instead of to the
 * world's capacity (the facade's think() runs every frame: PhysicsWorld.cpp:1356-1443; its listeners get one call per event, :1499-1520). */
int  sgp_world_event_counts(sgp_world* w, uint32_t counts_out[5]);
/* ---- static compound bodies ---------------------------------------------------
 * Replaces JPH::StaticCompoundShapeSettings::AddShape(position, rotation, shape, user data) x n + Create() as MeshBuilding.cpp:396-407
 * uses it for portals (the arch mesh + a thin box across the opening).  The compound is a STATIC body made of n child shapes, each with
 * a pose in the compound's frame; every child occupies its own body slot(s) (world pose = compound pose o child pose, all children share
 * the desc's material, layer and userdata), so the broad phase, narrow phase and queries need no notion of a sub-shape.  The id returned
 * (= the first child's slot) stands for the whole compound: pose setters, layer changes and removal act on every child; ray hits,
 * capsule-query contacts and contact events report this id, with the child index in `sub_shape` where the struct has one.  `base` gives
 * pose, layer, friction, restitution, sensor flag and userdata; its shape fields are ignored; motion_type must be SGP_MOTION_STATIC. */
typedef struct sgp_compound_child {
	int32_t shape_type;         /* SGP_SHAPE_SPHERE / BOX / CAPSULE / HULL / MESH */
	float   shape[4];           /* as sgp_body_desc::shape */
	float   pos[3];             /* of the child's body frame in the compound's frame */
	float   rot[4];
} sgp_compound_child;
#define SGP_MAX_COMPOUND_CHILDREN 64
int  sgp_body_add_compound(sgp_world* w, const sgp_body_desc* base, const sgp_compound_child* children, uint32_t num_children, uint32_t* id_out);
/* Number of children of compound `id` (0 = not a compound). */
int  sgp_body_compound_size(sgp_world* w, uint32_t id, uint32_t* num_children_out);

/* Body::GetUserData() of a live body (what JPH::BodyLockRead users read, PlayerPhysics.cpp:519-530); host-side lookup, no device access. */
int  sgp_body_get_userdata(sgp_world* w, uint32_t id, uint64_t* userdata_out);
/* getNumObjects (:1635-1638) */
int  sgp_world_num_bodies(sgp_world* w, uint32_t* n_out);
/* The counters PhysicsWorld::getDiagnostics prints (PhysicsWorld.cpp:1578-1604: JPH::BodyManager::BodyStats + the mesh count of getMemUsageStats). */
typedef struct sgp_body_counts {
	uint32_t num_bodies, max_bodies;
	uint32_t num_static, num_dynamic, num_kinematic;
	uint32_t num_active_dynamic, num_active_kinematic;
	uint32_t num_meshes;             /* live triangle-mesh shapes */
	uint32_t num_hulls;              /* live convex-hull shapes   */
	uint32_t reserved_;
	uint64_t shape_bytes;            /* device bytes held by the mesh and hull tables */
} sgp_body_counts;
int  sgp_world_body_counts(sgp_world* w, sgp_body_counts* out);

/* Debug / test view (not a facade entry point): the contact constraints of the last step, sorted by (a,b).
 * Record layout: {u32 a,b; i32 colour,np; f32 n[3], lam_n[4], lam_t1[4], lam_t2[4], bias[4]}. */
int  sgp_world_dump_constraints(sgp_world* w, void* out, uint32_t cap, uint32_t* n_out);

/* ---- queries --------------------------------------------------------------------------------- */
/* traceRay / traceRayAgainstCollidableObs / doesRayHitAnything (PhysicsWorld.cpp:1668-1725), batched. */
int  sgp_raycast(sgp_world* w, const sgp_ray* rays, uint32_t n, sgp_hit* hits_out);

/* ---- convex hull shapes ---------------------------------------------------------
 * Replaces JPH::ConvexHullShapeSettings(points).Create() (+ OffsetCenterOfMassShape / the principal-axes decomposition Jolt does
 * inside MassProperties) for dynamic meshes and vehicle bodies (gui_client/PhysicsWorld.cpp:735-1166 with is_dynamic,
 * CarPhysics.cpp:66-92, BikePhysics.cpp:76-112).  Up to 256 hull vertices / 512 faces / 768 edges (JPH::ConvexHullShape::cMaxPointsInHull; 32 / 60 before round 5); larger clouds are
 * reduced to their extreme points (every input point takes part, up to 100000).  Points must already carry the object's scale (ScaledShape is baked in).
 * The hull is stored in its BODY frame (origin = centre of mass, axes = principal axes of inertia).  `com` / `rot` give that
 * frame in the frame of the input points:  input point = com + rot * body point.  A caller that thinks in the points' frame
 * (object space) places the body at  pos_body = pos_obj + R_obj * com,  rot_body = rot_obj * rot. */
typedef struct sgp_hull_info {
	uint32_t hull_id;                 /* >= 1; goes into sgp_body_desc::shape[0] with shape_type = SGP_SHAPE_HULL                */
	uint32_t num_vertices, num_faces, num_edges;
	float com[3];  float rot[4];
	float volume;  float unit_inertia[3];     /* principal moments for density 1 (body inertia = unit_inertia * mass / volume)    */
	float aabb_min[3], aabb_max[3];           /* body frame                                                                      */
} sgp_hull_info;
int  sgp_hull_create(sgp_world* w, const float* points_xyz, uint32_t num_points, sgp_hull_info* info_out);
/* The same wrapped in JPH::OffsetCenterOfMassShapeSettings(com_offset, hull) (PhysicsWorld.cpp:1138-1153, CarPhysics.cpp:76-78,
 * BikePhysics.cpp:103-105): the body's centre of mass sits at hull centre of mass + com_offset (frame of the points). */
int  sgp_hull_create_com(sgp_world* w, const float* points_xyz, uint32_t num_points, const float com_offset[3], sgp_hull_info* info_out);
/* The last JPH::Ref<JPH::Shape> to the hull going away: its id becomes reusable.  SGP_ERR_REJECTED while a body still uses it. */
int  sgp_hull_destroy(sgp_world* w, uint32_t hull_id);

/* ---- static triangle meshes ---------------------------------------------------
 * Replaces JPH::MeshShapeSettings(vertices, triangles).Create() for static mesh objects and -- through a triangulated grid --
 * JPH::HeightFieldShapeSettings for the terrain (gui_client/PhysicsWorld.cpp:735-1166 with is_dynamic = false, :1020-1120;
 * TerrainSystem.cpp:1300).  Vertices must already carry the object's scale.  Front faces (counter-clockwise) collide, back
 * faces do not.  Mesh bodies are static, live in the mesh's own frame (no centre-of-mass shift) and take three body ids. */
typedef struct sgp_mesh_info {
	uint32_t mesh_id;                 /* >= 1; goes into sgp_body_desc::shape[0] with shape_type = SGP_SHAPE_MESH */
	uint32_t num_vertices, num_triangles, num_nodes;
	float aabb_min[3], aabb_max[3];
} sgp_mesh_info;
int  sgp_mesh_create(sgp_world* w, const float* vertices_xyz, uint32_t num_vertices, const uint32_t* indices, uint32_t num_triangles, sgp_mesh_info* info_out);
/* The last JPH::Ref<JPH::Shape> to the mesh going away: its id and its vertex / triangle / node storage become reusable (Substrata streams
 * static meshes in and out as the camera moves).  SGP_ERR_REJECTED while a body still uses it. */
/* The active-edge bits of a mesh's triangles in the caller's triangle order: bit k set = edge k (vertex k -> vertex k + 1) collides with its own normal;
 * a clear bit = an edge shared by two (nearly) coplanar triangles, or a concave one: a contact there takes the triangle's normal (JPH::MeshShape's active
 * edges with the 5 degree default the reference keeps, PhysicsWorld.cpp:1028-1060).  Diagnostics / tests. */
int  sgp_mesh_edge_flags(sgp_world* w, uint32_t mesh_id, uint8_t* flags_out, uint32_t cap);
int  sgp_mesh_destroy(sgp_world* w, uint32_t mesh_id);
/* The same with one user-data word per triangle (JPH::IndexedTriangle::mMaterialIndex / MeshShape::GetTriangleUserData: the reference
 * stores the batch's material index there, PhysicsWorld.cpp:1032-1060), reported by ray hits.  NULL = all 0. */
int  sgp_mesh_create_with_materials(sgp_world* w, const float* vertices_xyz, uint32_t num_vertices, const uint32_t* indices, uint32_t num_triangles,
                                    const uint32_t* triangle_materials, sgp_mesh_info* info_out);

/* ---- wheeled vehicles --------------------------------------------------------
 * Replaces JPH::VehicleConstraint + JPH::WheeledVehicleController + JPH::VehicleCollisionTesterCastSphere as CarPhysics
 * (and JPH::MotorcycleController + JPH::VehicleCollisionTesterCastCylinder as BikePhysics, gui_client/BikePhysics.cpp:97-230)
 * sets them up (gui_client/CarPhysics.cpp:62,94-231; script defaults gui_client/Scripting.cpp:315-346).  A vehicle is
 * attached to an existing dynamic body (the chassis; body origin = centre of mass).  Each step, before the forces:
 * one sphere cast per wheel, tyre slip -> friction, engine / clutch / gearbox / differential, brakes, anti-roll bars;
 * then 4 axis rows per wheel (suspension spring, max-up stop, longitudinal, lateral) are solved with the contact
 * constraints (before them in every iteration).  Field names follow JPH::WheelSettingsWV / VehicleEngineSettings /
 * VehicleTransmissionSettings / VehicleDifferentialSettings; sgp_default_vehicle_desc() fills Jolt's defaults. */
#define SGP_MAX_WHEELS 4
#define SGP_MAX_GEARS  8
#define SGP_VEHICLE_CONTROLLER_WHEELED     0   /* JPH::WheeledVehicleController (CarPhysics)                        */
#define SGP_VEHICLE_CONTROLLER_MOTORCYCLE  1   /* JPH::MotorcycleController (BikePhysics): + lean spring, lean steering limit */
#define SGP_VEHICLE_TESTER_SPHERE    0   /* JPH::VehicleCollisionTesterCastSphere of cast_radius (CarPhysics.cpp:62); radius 0: VehicleCollisionTesterRay */
#define SGP_VEHICLE_TESTER_CYLINDER  1   /* JPH::VehicleCollisionTesterCastCylinder(layer, inConvexRadiusFraction = 1) as BikePhysics.cpp:229 makes it: the wheel
                                          * itself is cast -- radius `radius`, width `width`, rounded by half its width -- from the attachment point over
                                          * suspension_max_length; no slope filter.  (Other fractions are not offered.) */
typedef struct sgp_wheel_desc {
	float position[3];            /* mPosition: suspension attachment point, chassis frame                       */
	float suspension_dir[3];      /* mSuspensionDirection (default (0,0,-1))                                      */
	float steering_axis[3];       /* mSteeringAxis (0,0,1)                                                        */
	float wheel_up[3];            /* mWheelUp (0,0,1)                                                             */
	float wheel_forward[3];       /* mWheelForward (0,1,0)                                                        */
	float suspension_min_length, suspension_max_length, suspension_preload;   /* 0.3, 0.5, 0                     */
	float spring_frequency, spring_damping;                                   /* 1.5 Hz, 0.5                     */
	float radius, width;                                                      /* 0.3, 0.1                        */
	float inertia, angular_damping;                                           /* 0.9 kg m^2, 0.2                 */
	float max_steer_angle, max_brake_torque, max_handbrake_torque;            /* 70 deg, 1500, 4000 N m          */
	float longitudinal_friction[3][2];   /* (slip ratio, friction): (0,0) (0.06,1.2) (0.2,1)                      */
	float lateral_friction[3][2];        /* (slip angle in degrees, friction): (0,0) (3,1.2) (20,1)               */
} sgp_wheel_desc;
typedef struct sgp_differential_desc { int32_t left_wheel, right_wheel; float differential_ratio, left_right_split, limited_slip_ratio, engine_torque_ratio; } sgp_differential_desc;
typedef struct sgp_anti_roll_bar_desc { int32_t left_wheel, right_wheel; float stiffness; } sgp_anti_roll_bar_desc;
typedef struct sgp_vehicle_desc {
	uint32_t body;                /* chassis body id                                                              */
	uint32_t num_wheels;          /* 1..SGP_MAX_WHEELS                                                            */
	sgp_wheel_desc wheels[SGP_MAX_WHEELS];
	float up[3], forward[3];      /* VehicleConstraintSettings::mUp / mForward, chassis frame                      */
	float cast_radius;            /* VehicleCollisionTesterCastSphere radius (CarPhysics: 0.5 * wheel width); 0 = ray */
	float max_slope_angle;        /* hits steeper than this against world +z are ignored (80 deg)                  */
	float engine_max_torque, engine_min_rpm, engine_max_rpm, engine_inertia, engine_angular_damping;   /* 500, 1000, 6000, 0.5, 0.2 */
	float engine_torque_curve[3][2];     /* (rpm / max rpm, torque fraction): (0,0.8) (0.66,1) (1,0.8)             */
	uint32_t num_gears, num_reverse_gears;
	float gear_ratios[SGP_MAX_GEARS], reverse_gear_ratios[SGP_MAX_GEARS];    /* 2.66 1.78 1.3 1.0 0.74 / -2.9     */
	float switch_time, clutch_release_time, switch_latency, shift_up_rpm, shift_down_rpm, clutch_strength;   /* .5 .3 .5 4000 2000 10 */
	uint32_t num_differentials;
	sgp_differential_desc differentials[2];
	float differential_limited_slip_ratio;   /* 1.4 */
	uint32_t num_anti_roll_bars;
	sgp_anti_roll_bar_desc anti_roll_bars[2];
	/* JPH::MotorcycleControllerSettings (BikePhysics.cpp:197-205); ignored for SGP_VEHICLE_CONTROLLER_WHEELED */
	uint32_t controller_type;                /* SGP_VEHICLE_CONTROLLER_*                                            */
	float max_lean_angle;                    /* mMaxLeanAngle 45 deg                                               */
	float lean_spring_constant, lean_spring_damping;                       /* 5000, 1000                            */
	float lean_spring_integration_coefficient, lean_spring_integration_decay;   /* 0, 4                              */
	float lean_smoothing_factor;             /* 0.8                                                                */
	uint32_t lean_steering_limit;            /* mEnableLeanSteeringLimit (1)                                       */
	uint32_t collision_tester;               /* SGP_VEHICLE_TESTER_* (0: the sphere / ray of cast_radius)          */
} sgp_vehicle_desc;
/* WheeledVehicleController::SetDriverInput(forward, right, brake, hand brake) (CarPhysics.cpp:366-367) */
typedef struct sgp_vehicle_input { float forward, right, brake, hand_brake; } sgp_vehicle_input;
/* What CarPhysics reads back from JPH::Wheel (CarPhysics.cpp:405-470) and the controller (BikePhysics.cpp:707) */
typedef struct sgp_wheel_state {
	float suspension_length, steer_angle, rotation_angle, angular_velocity;
	int32_t has_contact; uint32_t contact_body;
	float contact_position[3], contact_normal[3], contact_longitudinal[3], contact_lateral[3], contact_point_velocity[3];
	float suspension_lambda, longitudinal_lambda, lateral_lambda;
	float longitudinal_slip, lateral_slip;
} sgp_wheel_state;
typedef struct sgp_vehicle_state {
	sgp_wheel_state wheels[SGP_MAX_WHEELS];
	float engine_rpm; int32_t current_gear; float clutch_friction; int32_t active;
} sgp_vehicle_state;
void sgp_default_vehicle_desc(sgp_vehicle_desc* d);         /* Jolt defaults + CarPhysics' 4-wheel FWD layout, Scripting.cpp defaults */
int  sgp_vehicle_create(sgp_world* w, const sgp_vehicle_desc* d, uint32_t* vehicle_id_out);   /* AddConstraint + AddStepListener, CarPhysics.cpp:224-226 */
int  sgp_vehicle_destroy(sgp_world* w, uint32_t vehicle_id);                                  /* RemoveConstraint / RemoveStepListener, :258-262 */
int  sgp_vehicle_set_input(sgp_world* w, uint32_t vehicle_id, const sgp_vehicle_input* in);
int  sgp_vehicle_set_inputs(sgp_world* w, uint32_t first_vehicle_id, uint32_t n, const sgp_vehicle_input* in);
int  sgp_vehicle_get_state(sgp_world* w, uint32_t vehicle_id, sgp_vehicle_state* out);
int  sgp_vehicle_get_states(sgp_world* w, uint32_t first_vehicle_id, uint32_t n, sgp_vehicle_state* out);
/* MotorcycleController::EnableLeanController (BikePhysics.cpp:493,617) */
int  sgp_vehicle_enable_lean_controller(sgp_world* w, uint32_t vehicle_id, int enabled);
/* vehicleSummoned(): GetEngine().SetCurrentRPM / Wheel::SetAngularVelocity (CarPhysics.cpp:266-272) */
int  sgp_vehicle_reset_drivetrain(sgp_world* w, uint32_t vehicle_id, float engine_rpm, float wheel_angular_velocity);

/* ---- shape queries for the character controller ----------------------------------------------------
 * What JPH::CharacterVirtual asks the world every update (PlayerPhysics.cpp:64-90,258-353,477-481): every body within reach of
 * the character's capsule (CollideShape with a maximum separation) and a swept test of its path.  The controller itself (plane
 * constraints, sliding, ground state, stairs) is host code on top of these two batched queries (shim/Jolt/JoltCharacterLite.h). */
typedef struct sgp_capsule_query {
	float    pos[3];            /* centre of the capsule                                                            */
	float    rot[4];            /* the capsule axis is the local z axis                                             */
	float    radius, half_height;
	float    max_separation;    /* report surfaces closer than this (predictive contact distance + character padding) */
	uint32_t ignore_id;         /* JPH::IgnoreSingleBodyFilter (PlayerPhysics.cpp:477)                             */
	uint32_t collidable_only;   /* PlayerPhysicsObjectLayerFilter (PlayerPhysics.cpp:240-249)                      */
	float    movement[3];       /* CollideShapeSettings::mActiveEdgeMovementDirection (CharacterVirtual passes the direction it moves in); only its direction matters */
	uint32_t active_edges;      /* 1: EActiveEdgeMode::CollideOnlyWithActive -- a hit on an inactive edge of a mesh takes the triangle's normal unless the
	                               movement runs against the triangle more steeply along the found one (what CharacterVirtual asks for); 0: every edge collides with its own normal */
} sgp_capsule_query;
typedef struct sgp_query_contact {
	uint32_t query;             /* index of the query                                                              */
	uint32_t body;
	float    point[3];          /* on the body                                                                     */
	float    normal[3];         /* from the body towards the capsule                                               */
	float    distance;          /* separation along the normal; negative = penetration depth                       */
	float    point_velocity[3]; /* velocity of the body at `point`                                                 */
	uint32_t motion_type;       /* SGP_MOTION_*                                                                    */
	uint32_t is_sensor;
	float    inv_mass;          /* 0 unless dynamic                                                                */
	uint32_t sub_shape;         /* compound bodies: index of the touched child (`body` is the compound's id); 0 otherwise
	                               -> JPH::SubShapeID of CharacterContactListener::OnContactAdded (GUIClient.cpp:6484-6486)   */
	uint64_t userdata;
} sgp_query_contact;
/* Contacts come back sorted by (query, body, point). n_out may exceed cap (then only the first cap are written). */
int  sgp_collide_capsules(sgp_world* w, const sgp_capsule_query* queries, uint32_t n, sgp_query_contact* out, uint32_t cap, uint32_t* n_out);
/* Sphere casts (rays with thickness): hit.t = distance travelled by the centre until first touch, hit.normal at the touch point. */
int  sgp_spherecast(sgp_world* w, const sgp_ray* rays, const float* radii, uint32_t n, sgp_hit* hits_out);

/* ---- multi-GPU tiles: ghost bodies are ordinary kinematic-like bodies owned elsewhere ---- */
/* Pack the ghost record of every owned body whose AABB, inflated by `margin`, crosses outside [lo,hi). */
typedef struct sgp_ghost_record {
	float    pos[3];  float rot[4];  float lin_vel[3];  float ang_vel[3];
	int32_t  shape_type;  float shape[4];
	float    mass;  float friction;  float restitution;
	uint32_t motion_type;
	uint64_t global_id;          /* owner's local body id | owner's rank << 40 */
	/* the rest of the body's description: what an ownership migration needs to re-create the body as it was (128 bytes in all) */
	uint64_t userdata;           /* mUserData = the caller's PhysicsObject*: events and ray hits of the new owner keep reporting the same object */
	float    gravity_factor;  float linear_damping;  float angular_damping;
	uint32_t flags;              /* SGP_GHOST_FLAG_*: layer in bits 0-1, then is_sensor, allow_sleeping, use_zero_linear_drag */
	uint32_t _pad[2];
} sgp_ghost_record;
#define SGP_GHOST_FLAG_LAYER_MASK   0x3u
#define SGP_GHOST_FLAG_SENSOR       (1u << 2)
#define SGP_GHOST_FLAG_ALLOW_SLEEP  (1u << 3)
#define SGP_GHOST_FLAG_ZERO_DRAG    (1u << 4)
#define SGP_GHOST_FLAG_CHASSIS      (1u << 5)   /* the body carries a live vehicle: it never changes owner (the vehicle record -- engine, gearbox, wheel state -- lives with the tile that created it); its neighbours see it as a ghost like any other body */
int  sgp_world_export_boundary(sgp_world* w, const float lo[3], const float hi[3], float margin,
                               sgp_ghost_record* out, uint32_t cap, uint32_t* n_out);
/* Replace this world's ghost set with `n` records (bodies simulated as velocity-driven, infinite mass). */
int  sgp_world_import_ghosts(sgp_world* w, const sgp_ghost_record* in, uint32_t n);
/* Host-side routing of one tile's exported records (pure functions, no world, no device): the per-step bookkeeping of the exchange
 * in substrata_amd/tiles.py, which numpy on 100-byte records is too slow for.
 * sgp_tiles_route: `boxes` holds n_tiles x (lo xyz, hi xyz).  A record goes to every OTHER tile whose region grown by `pad` contains its
 * centre; `send_out` receives the records grouped by destination in rank order (`send_counts[r]` each; a record can appear under several
 * destinations; cap = capacity of send_out in records, exceeded -> SGP_ERR_CAPACITY).  Owned DYNAMIC bodies whose centre has left
 * boxes[my_rank] emigrate: their records carry SGP_GHOST_TAKE_OWNERSHIP in motion_type and the low 32 bits of their global ids (the
 * local body ids) are listed in `emigrant_ids`.  `rank_tag` is OR-ed into bits 40.. of every global_id. */
#define SGP_GHOST_TAKE_OWNERSHIP 0x100u
int  sgp_tiles_route(const sgp_ghost_record* recs, uint32_t n, uint32_t my_rank, const float* boxes, uint32_t n_tiles, float pad,
                     sgp_ghost_record* send_out, uint32_t cap, uint32_t* send_counts,
                     uint32_t* emigrant_ids, uint32_t emigrant_cap, uint32_t* n_emigrants);
/* sgp_tiles_split: what arrived at a tile -> ghosts (records without the flag) and immigrants (flagged records whose centre lies in
 * [lo,hi); flagged records addressed to another tile are dropped).  Both outputs need room for n records. */
int  sgp_tiles_split(const sgp_ghost_record* in, uint32_t n, const float lo[3], const float hi[3],
                     sgp_ghost_record* ghosts_out, uint32_t* n_ghosts, sgp_ghost_record* immigrants_out, uint32_t* n_immigrants);

/* ---- the exchange itself, below the C ABI (what a C++ host with one world per GPU calls once per sub-step before sgp_world_step) ----
 * The routing runs on the device (one record per (boundary body, destination tile), segmented by destination), the per-destination
 * counts are all-gathered over RCCL and the records travel device to device with grouped ncclSend / ncclRecv over xGMI; the receiving
 * tile refreshes its ghosts with a kernel while the ghost set is unchanged, and hands the records to the host (which owns the body
 * slots) only when ghosts appear / disappear or bodies migrate.  Ownership migrates with the body's full description (sgp_ghost_record).
 *
 *   rank 0:  sgp_tiles_unique_id(id)  -> the application broadcasts the 128 bytes (MPI, a socket, torch.distributed ...)
 *   all:     sgp_tiles_create(world, rank, n_tiles, boxes, margin, radius_pad, id, &tiles)        (collective: ncclCommInitRank)
 *   per step: sgp_tiles_exchange(tiles); sgp_world_step(world, dt);
 *
 * Tiles that live in ONE process (several GPUs driven by one thread, or a single-GPU dry run) are created with unique_id = NULL and
 * exchanged together with sgp_tiles_exchange_group (device-to-device copies, no communicator).
 * boxes: n_tiles x (lo xyz, hi xyz).  margin: a body whose AABB grown by it leaves the tile is exported; a record goes to every other
 * tile whose region grown by margin + radius_pad contains the body's centre (radius_pad >= the largest bounding radius). */
typedef struct sgp_tiles sgp_tiles;
#define SGP_TILES_UNIQUE_ID_BYTES 128
typedef struct sgp_tiles_stats {
	uint32_t exported;       /* records this tile produced in the last exchange (a body counts once per destination)     */
	uint32_t sent;
	uint32_t received;       /* records that arrived                                                                     */
	uint32_t ghosts;         /* ... of which ghosts                                                                      */
	uint32_t emigrated, immigrated;
	uint32_t fast_imports, slow_imports;     /* cumulative: exchanges that found the ghost set unchanged (the host saw 16 bytes per record) / whose 128-byte records had to come to the host (round 6: only sets with hull or mesh records; a changed set of primitives costs the host 32 bytes per record, and the newcomers are created on the device from the records: device_creates); the ghosts' POSES go from the received records to the bodies on the device either way */
	uint32_t route_retries;  /* exchanges in which some rank had to grow a send / emigrant / receive buffer: every rank routes again and the counts are gathered once more */
	uint32_t comm_ranks;     /* ranks ncclCommCount reports for this tile's communicator (0: no communicator, e.g. tiles of one process) */
	uint32_t exchanges;      /* sgp_tiles_exchange calls so far                                                           */
	float    comm_init_ms;   /* wall time of ncclCommInitRank                                                             */
	float    last_exchange_ms, total_exchange_ms;    /* host wall time of sgp_tiles_exchange: the last call, all calls    */
	uint32_t rebalances;     /* sgp_tiles_rebalance calls that moved this tile's region                                  */
	uint32_t device_creates; /* cumulative: ghosts and immigrants created on the device straight from the received records (no record to the host, no create command back) */
} sgp_tiles_stats;
#define SGP_MIGRATION_OUT 0
#define SGP_MIGRATION_IN  1
typedef struct sgp_migration {
	uint64_t userdata;       /* the body's user data (its PhysicsObject*): how a caller finds the object whose id changed          */
	uint32_t old_id;         /* OUT: the id it had in this world (now free).  IN: the id it had in its previous owner's world      */
	uint32_t new_id;         /* IN: its id in this world.  OUT: SGP_INVALID_ID                                                     */
	uint32_t direction;      /* SGP_MIGRATION_OUT / SGP_MIGRATION_IN                                                               */
	uint32_t peer;           /* IN: rank of the previous owner                                                                     */
} sgp_migration;
int  sgp_tiles_unique_id(uint8_t id_out[SGP_TILES_UNIQUE_ID_BYTES]);
int  sgp_tiles_create(sgp_world* w, uint32_t rank, uint32_t n_tiles, const float* boxes, float margin, float radius_pad,
                      const uint8_t* unique_id, sgp_tiles** tiles_out);
int  sgp_tiles_destroy(sgp_tiles* t);
int  sgp_tiles_exchange(sgp_tiles* t);
int  sgp_tiles_exchange_group(sgp_tiles** tiles, uint32_t n_tiles);
int  sgp_tiles_get_stats(sgp_tiles* t, sgp_tiles_stats* out);
/* Re-tiling by body count (collective; round 4).  A static split of a scene that moves leaves tiles without work -- BASELINE config 4 is a tower that
 * falls out of its upper tiles.  The grid gx x gy x gz (gx gy gz = n_tiles, tile = ix + gx (iy + gy iz), at most 4 per axis) keeps its topology; its
 * split planes move to the quantiles of where the OWNED dynamic bodies are (x planes from all bodies, the y planes of an x slab from that slab's, the z
 * planes of an (x, y) column from that column's).  Every tile computes the same planes from the same all-gathered counts; bodies then change owner
 * through the ordinary migration of the following exchanges.  sgp_tiles_rebalance: one tile per process, over the communicator;
 * sgp_tiles_rebalance_group: all tiles of one process.  sgp_tiles_get_boxes: the regions now in force (n_tiles x (lo xyz, hi xyz)).
 * by_contacts != 0: a body counts 1 + the contact constraints it was in during the last step (a tile's work is its constraints more than its bodies). */
int  sgp_tiles_rebalance(sgp_tiles* t, uint32_t gx, uint32_t gy, uint32_t gz, int by_contacts);
int  sgp_tiles_rebalance_group(sgp_tiles** tiles, uint32_t n_tiles, uint32_t gx, uint32_t gy, uint32_t gz, int by_contacts);
int  sgp_tiles_get_boxes(sgp_tiles* t, float* boxes_out);
/* Ownership changes since the last drain (both directions), oldest first; n_out = how many were pending. */
int  sgp_tiles_drain_migrations(sgp_tiles* t, sgp_migration* out, uint32_t cap, uint32_t* n_out);

/* ---- device-resident bulk access (bench / torch plumbing; pointers are HIP device pointers) ---- */
/* Raw views of the body records, valid until the world is destroyed; record i = FOUR float4 (64 bytes) at [4 i] .. [4 i + 3], of which the first two are
 * the caller's: which = 0 pose: (position xyz, inverse mass) (rotation quaternion xyzw), then two float4 of body properties; 1 velocity: (linear velocity
 * xyz, -) (angular velocity xyz, -), then two float4 that belong to the solver (the step's world-space inverse inertia). */
int  sgp_world_device_array(sgp_world* w, int which, void** dev_ptr_out, uint32_t* count_out);
/* hipStream_t the world launches on (as void*). */
int  sgp_world_stream(sgp_world* w, void** stream_out);

#ifdef __cplusplus
}
#endif
#endif /* SGP_H */
