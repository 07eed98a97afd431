/*
 * sgo_oracle.c -- THE ORACLE.  Test infrastructure only: nothing under substrata_amd/ may include,
 * link or call this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * What it restates.  PhysicsWorld::think(dt)  (/root/reference/gui_client/PhysicsWorld.cpp:1356-1443):
 *     physics_system->Update((float)dt, cCollisionSteps = 1, ...)          :1359-1363
 *     then the water-buoyancy sweep over activated dynamic bodies           :1367-1442
 * The arithmetic of Update() lives in JoltPhysics v5.3.0 (scripts/get_libs.rb:29-40), which is NOT in
 * /root/reference and not on this machine.  This file therefore restates Jolt's published step
 * (PhysicsSystem::Update: apply forces -> find collisions -> contact constraints with a contact cache ->
 * sequential-impulse velocity solve -> integrate -> position solve -> island sleeping) from upstream
 * knowledge, under exactly the configuration Substrata imposes (all citable):
 *     gravity (0,0,-9.81) :520, one collision step :1359, layer matrix :151-189, body settings :1229-1243,
 *     unit sphere r 0.5 / unit cube half 0.5 :1221-1255, buoyancy constants :1384-1410,
 *     Jolt default PhysicsSettings (Substrata never calls SetPhysicsSettings).
 *
 * parity unpinned: the reference holds no golden vectors, KATs or fixtures for this path
 * (PhysicsWorld::test(), :1754-1825, asserts nothing about dynamics) and Jolt cannot be built here, so this
 * oracle is pinned only by the analytic known-answer tests in tests/test_oracle_kat.py.
 *
 * Ordering contract shared with the device implementation (so that results are comparable to fp32 rounding):
 *   - the set of body pairs / manifolds is order independent;
 *   - contact constraints are solved colour by colour; colours come from the deterministic round-based
 *     greedy colouring below (priority = sgp_mix64(pair key)); constraints of one colour share no movable
 *     body, so their relative order cannot matter.  This is a legal sequential-impulse order (Jolt itself
 *     batches large islands into non-conflicting splits, LargeIslandSplitter).
 */
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "../include/sgp.h"
#include "sgo_collide.h"
#include "sgo_hull_build.h"
#include "sgo_vehicle.h"
#include "sgo_mesh.h"

#define SGO_API __attribute__((visibility("default")))
#define SGO_MAX_COLOURS 64
#define SGO_OVERFLOW_COLOUR 63

typedef struct {
	v3 pos; quat rot; v3 linv, angv; v3 force, torque;
	float inv_mass; v3 inv_inertia;
	int shape_type; float shape[4];
	const struct sgo_mesh_s* mesh;    /* SGP_SHAPE_MESH: the triangles (mesh frame = body frame) */
	int is_alias;                     /* internal second / third slot of a mesh body: carries contact manifolds only */
	uint32_t comp_root, comp_child;   /* child of a static compound body: the compound's id (= first child's slot) and the child's index; root = SGP_INVALID_ID otherwise */
	float comp_local_pos[3], comp_local_rot[4];   /* ... and its pose in the compound's frame */
	float comp_pos[3], comp_rot[4];   /* (on the first child) the compound's own pose */
	uint32_t comp_n;                  /* (on the first child) number of children */
	const sgo_hull* hull;             /* SGP_SHAPE_HULL: the shape; SGP_SHAPE_BOX: the +-1 cube template (for box - hull pairs) */
	int motion, layer;
	float friction, restitution, gravity_factor, lin_damp, ang_damp, mass;
	int is_sensor, allow_sleep, zero_lin_drag;
	uint64_t userdata;
	int alive, active;
	v3 aabb_min, aabb_max;
	v3 sleep_c[3]; float sleep_r[3]; float sleep_timer;
	uint32_t slot_gen;                 /* how often this SLOT has been given to a new body (7 bits are used): part of every label made from its id */
	uint32_t sleep_label;              /* SGO_LABEL(id, generation of that slot when the label was made): the island this body fell asleep with (the member with the lowest uf_prio; its own id from creation): bodies that share it wake together */
	int underwater; float submerged;
	/* per-step scratch */
	uint64_t colour_mask; uint64_t claim[2];
	int chassis;                       /* chassis of a live vehicle (set by colour_constraints): its contacts do not take colour 0 */
	int movable_prev, movable_cur;   /* movable (dynamic and awake) when the previous / this step coloured its constraints */
	int island; int can_sleep;
	v3 linv_pre;                       /* linear velocity at the start of the step (before the forces): the active-edge movement hint */
	int awake_step;                  /* awake (active and not static) in the step being taken, once its collision detection has woken what it wakes */
	int fresh;                       /* was cache_invalid when the step being taken began */
	int cache_invalid;               /* created or reshaped since the last step: its pairs do not reuse cached manifolds (Body::InvalidateContactCache) */
} sgo_body;

typedef struct {
	v3 r1, r2;           /* world offsets from the centres of mass to the contact mid point */
	v3 local1, local2;   /* contact point on 1 / 2 in the body frames (cache matching, position solve) */
	float bias;
	float eff_n, eff_t1, eff_t2;
	float lam_n, lam_t1, lam_t2;
} sgo_point;

typedef struct {
	uint32_t a, b;       /* a < b */
	uint64_t key, prio;
	v3 n, t1, t2;
	float friction;
	int np;
	int colour;
	int persisted;
	/* body-pair contact cache (ContactConstraintManager::GetContactsFromCache): pose of body 2 relative to body 1 and the normal in body 2's
	   frame WHEN THE MANIFOLD WAS COMPUTED; a reused manifold keeps them, so slow drift ends the reuse */
	v3 dpos; quat drot; v3 nloc2; int reused;
	int carried;         /* not of the last step: the contact of a pair both of whose bodies have been asleep (or static) since -- kept in the cache, never solved */
	sgo_point pt[4];
} sgo_constraint;

typedef struct { uint32_t a, b; } sgo_pair;

typedef struct sgo_mesh_s { uint32_t nv, nt; v3* verts; uint32_t* tris; uint32_t* mats; unsigned char* edges; v3 aabb_min, aabb_max; float bound_radius; } sgo_mesh;

#define SGO_LABEL(id, gen) ((uint32_t)(id) | ((uint32_t)(gen) << 25))
#define SGO_LABEL_SLOT(l) ((l) & 0x1FFFFFFu)
/* a label is current while the slot it names still holds the body (generation) it was made from */
#define SGO_LABEL_CURRENT(w, l) (SGO_LABEL_SLOT(l) < (w)->cap && ((l) >> 25) == ((w)->bodies[SGO_LABEL_SLOT(l)].slot_gen & 0x7Fu))
typedef struct sgo_world {
	sgp_world_desc desc;
	sgp_settings st;
	v3 gravity;
	uint32_t cap, high;          /* capacity, high-water slot count */
	sgo_body* bodies;
	uint32_t* free_list; uint32_t n_free;
	uint32_t* free_triples; uint32_t n_free_triples;      /* first slot of freed (mesh body + 2 alias) triples: the next mesh body reuses one */
	uint32_t* free_mesh_ids; uint32_t n_free_mesh_ids; uint32_t* free_hull_ids; uint32_t n_free_hull_ids;      /* ids of destroyed shapes (reused last-in first-out) */
	uint32_t n_alive;
	/* pairs / constraints of the current and previous step */
	sgo_pair* pairs; uint32_t n_pairs, cap_pairs;
	sgo_constraint* cons; uint32_t n_cons, cap_cons;
	sgo_constraint* prev; uint32_t n_prev, cap_prev;
	uint64_t* prev_keys_sorted; uint32_t* prev_idx_sorted;
	uint32_t* order;            /* constraint indices sorted by (colour, prio) */
	int water_enabled; float water_z;
	int contact_events;
	sgp_step_stats stats;
	/* events */
	sgp_body_event* ev_act; uint32_t n_act, cap_act;
	sgp_body_event* ev_deact; uint32_t n_deact, cap_deact;
	sgp_body_event* ev_water; uint32_t n_water, cap_water;
	sgp_contact_event* ev_added; uint32_t n_added, cap_added;
	sgp_contact_event* ev_pers; uint32_t n_pers, cap_pers;
	/* broad-phase scratch */
	uint64_t* cell_keys; uint32_t* cell_idx; uint32_t* large; uint32_t n_large;
	struct keyidx_s* bp_ki; uint32_t bp_n_small; float bp_cell;      /* the step's cell-sorted small bodies (kept until the next broad phase: find_contacts pairs the bodies it wakes) */
	/* multi-tile ghosts: global id -> local id, kept sorted by global id */
	uint64_t* ghost_gid; uint32_t* ghost_lid; uint32_t n_ghosts;
	int* is_ghost;
	uint64_t tot_act, tot_deact, rep_act, rep_deact;   /* running totals of (de)activation events / totals already reported in stats */
	sgo_mesh** meshes; uint32_t n_meshes, cap_meshes;      /* static triangle meshes; id 0 unused */
	/* convex hull shapes (sgo_hull.h): stable pointers, hull 0 = the +-1 cube template */
	sgo_hull** hulls; uint32_t n_hulls, cap_hulls;
	/* wheeled vehicles (sgo_vehicle.h) */
	sgo_vehicle* vehicles; uint32_t n_vehicles, cap_vehicles;
} sgo_world;

/* ------------------------------------------------------------------------------------------------ */
/* defaults (Jolt v5.3.0 PhysicsSettings / BodyCreationSettings; UNVERIFIED: upstream)                  */

SGO_API void sgo_default_settings(sgp_settings* s)
{
	/* Jolt v5.3.0 Jolt/Physics/PhysicsSettings.h, restated from memory: Jolt's sources are not in /root/reference (scripts/get_libs.rb:29-40
	   fetches them at build time), so none of these could be checked here.  oracle/jolt_ref/oracle_jolt.cpp prints the real values when it
	   can be built. */
	s->num_velocity_steps = 10;                                      /* mNumVelocitySteps             UNVERIFIED: upstream */
	s->num_position_steps = 2;                                       /* mNumPositionSteps             UNVERIFIED: upstream */
	s->baumgarte = 0.2f;                                             /* mBaumgarte                    UNVERIFIED: upstream */
	s->penetration_slop = 0.02f;                                     /* mPenetrationSlop              UNVERIFIED: upstream */
	s->speculative_contact_distance = 0.02f;                         /* mSpeculativeContactDistance   UNVERIFIED: upstream */
	s->min_velocity_for_restitution = 1.0f;                          /* mMinVelocityForRestitution    UNVERIFIED: upstream */
	s->max_penetration_distance = 0.2f;                              /* mMaxPenetrationDistance       UNVERIFIED: upstream */
	s->time_before_sleep = 0.5f;                                     /* mTimeBeforeSleep              UNVERIFIED: upstream */
	s->point_velocity_sleep_threshold = 0.03f;                       /* mPointVelocitySleepThreshold  UNVERIFIED: upstream */
	s->contact_point_preserve_lambda_max_dist_sq = 0.01f * 0.01f;    /* mContactPointPreserveLambdaMaxDistSq  UNVERIFIED: upstream */
	s->max_linear_velocity = 500.0f;                                 /* BodyCreationSettings::mMaxLinearVelocity   UNVERIFIED: upstream */
	s->max_angular_velocity = 0.25f * 3.14159265358979323846f * 60.0f; /* BodyCreationSettings::mMaxAngularVelocity  UNVERIFIED: upstream */
	s->allow_sleeping = 1;                                           /* mAllowSleeping                UNVERIFIED: upstream */
	s->warm_start = 1;                                               /* mConstraintWarmStart          UNVERIFIED: upstream */
	s->use_body_pair_contact_cache = 1;                              /* mUseBodyPairContactCache      UNVERIFIED: upstream */
	s->body_pair_cache_max_delta_position_sq = 0.001f * 0.001f;      /* mBodyPairCacheMaxDeltaPositionSq         UNVERIFIED: upstream */
	s->body_pair_cache_cos_max_delta_rotation_div2 = 0.99984769515639123915701155881391f;   /* mBodyPairCacheCosMaxDeltaRotationDiv2 = cos(2 deg / 2)   UNVERIFIED: upstream */
}

SGO_API void sgo_default_world_desc(sgp_world_desc* d)
{
	memset(d, 0, sizeof(*d));
	d->max_bodies = 65536;                 /* PhysicsWorld.cpp:492 */
	d->gravity[0] = 0.0f; d->gravity[1] = 0.0f; d->gravity[2] = -9.81f; /* :520 */
	d->large_body_radius = 4.0f;
	sgo_default_settings(&d->settings);
}

SGO_API void sgo_default_body_desc(sgp_body_desc* d)
{
	memset(d, 0, sizeof(*d));
	d->rot[3] = 1.0f;
	d->shape_type = SGP_SHAPE_BOX;
	d->shape[0] = d->shape[1] = d->shape[2] = 0.5f;   /* unit cube, PhysicsWorld.cpp:1249 */
	d->motion_type = SGP_MOTION_STATIC;                /* PhysicsObject.cpp:28 */
	d->layer = SGP_LAYER_NON_MOVING;
	d->mass = 100.0f; d->friction = 0.5f; d->restitution = 0.3f; /* PhysicsObject.cpp:36-38 */
	d->gravity_factor = 1.0f;                          /* BodyCreationSettings::mGravityFactor          UNVERIFIED: upstream */
	d->linear_damping = 0.05f; d->angular_damping = 0.05f;   /* mLinearDamping, mAngularDamping            UNVERIFIED: upstream */
	d->allow_sleeping = 1;                             /* mAllowSleeping                                UNVERIFIED: upstream */
}

/* Optional multi-core mode for the cpu_baseline timing (bench.py): the loops over bodies, pairs and the constraints of one
   colour are order independent (see the ordering contract above), so they are plain OpenMP parallel-for loops and give the
   same bits as the single-thread run.  Tests run with 1 thread. */
static int g_threads = 1;
SGO_API int sgo_set_threads(int n)
{
#ifdef _OPENMP
	if (n < 1) n = 1;
	g_threads = n;
	omp_set_num_threads(n);
	return n;
#else
	(void)n; return 1;
#endif
}
SGO_API int sgo_get_threads(void) { return g_threads; }

/* ------------------------------------------------------------------------------------------------ */
/* layer matrix: MyObjectLayerPairFilter, PhysicsWorld.cpp:160-189                                     */

static int layers_collide(int l1, int l2)
{
	switch (l1) {
	case SGP_LAYER_NON_MOVING: return l2 == SGP_LAYER_MOVING;
	case SGP_LAYER_MOVING: return l2 != SGP_LAYER_NON_MOVING_NON_COLLIDABLE && l2 != SGP_LAYER_MOVING_NON_COLLIDABLE;
	default: return 0;
	}
}
SGO_API int sgo_layers_collide(int l1, int l2) { return layers_collide(l1, l2); }

/* ------------------------------------------------------------------------------------------------ */
/* mass properties (Jolt Shape::GetMassProperties scaled to the overridden mass, CalculateInertia)      */

static void mass_properties(int type, const float* p, const sgo_hull* hull, float mass, float* inv_mass, v3* inv_inertia)
{
	v3 I;
	if (type == SGP_SHAPE_HULL) {
		const float density = mass / hull->volume;
		I = v3_scale(hull->unit_inertia, density);
	} else if (type == SGP_SHAPE_SPHERE) {
		const float i = 0.4f * mass * p[0] * p[0];
		I = V3(i, i, i);
	} else if (type == SGP_SHAPE_BOX) {
		const float sx = 2.0f * p[0], sy = 2.0f * p[1], sz = 2.0f * p[2];
		const float k = mass / 12.0f;
		I = V3(k * (sy * sy + sz * sz), k * (sx * sx + sz * sz), k * (sx * sx + sy * sy));
	} else {
		/* capsule along z: cylinder (height H = 2 hh) + two hemispheres, uniform density */
		const float r = p[0], H = 2.0f * p[1];
		const float vc = 3.14159265358979323846f * r * r * H;
		const float vs = (4.0f / 3.0f) * 3.14159265358979323846f * r * r * r;
		const float mc = mass * vc / (vc + vs), ms = mass * vs / (vc + vs);
		const float iz = 0.5f * mc * r * r + 0.4f * ms * r * r;
		const float ix = mc * (3.0f * r * r + H * H) / 12.0f
		               + ms * (0.4f * r * r + 0.25f * H * H + 0.375f * H * r);
		I = V3(ix, ix, iz);
	}
	*inv_mass = 1.0f / mass;
	*inv_inertia = V3(1.0f / I.x, 1.0f / I.y, 1.0f / I.z);
}

static float shape_volume_h(int type, const float* p, const sgo_hull* hull);
static float shape_volume(int type, const float* p)
{
	if (type == SGP_SHAPE_SPHERE) return (4.0f / 3.0f) * 3.14159265358979323846f * p[0] * p[0] * p[0];
	if (type == SGP_SHAPE_BOX) return 8.0f * p[0] * p[1] * p[2];
	return 3.14159265358979323846f * p[0] * p[0] * (2.0f * p[1]) + (4.0f / 3.0f) * 3.14159265358979323846f * p[0] * p[0] * p[0];
}

/* Local-space half extents of the shape's AABB (Shape::GetLocalBounds). */
static float shape_volume_h(int type, const float* p, const sgo_hull* hull) { return type == SGP_SHAPE_MESH ? 0.0f : (type == SGP_SHAPE_HULL ? hull->volume : shape_volume(type, p)); }

static v3 shape_local_half_h(int type, const float* p, const sgo_hull* hull);
static v3 shape_local_half(int type, const float* p)
{
	if (type == SGP_SHAPE_SPHERE) return V3(p[0], p[0], p[0]);
	if (type == SGP_SHAPE_BOX) return V3(p[0], p[1], p[2]);
	return V3(p[0], p[0], p[1] + p[0]);
}

static v3 shape_local_half_h(int type, const float* p, const sgo_hull* hull)
{
	if (type == SGP_SHAPE_MESH) return V3(1.0f, 1.0f, 1.0f);      /* (static: never asked for sleep points) */
	if (type == SGP_SHAPE_HULL) return v3_max(v3_abs(hull->aabb_min), v3_abs(hull->aabb_max));
	return shape_local_half(type, p);
}

static float shape_bounding_radius(int type, const float* p)
{
	if (type == SGP_SHAPE_SPHERE) return p[0];
	if (type == SGP_SHAPE_BOX) return sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
	return p[0] + p[1];
}

/* mesh bodies always count as large (tested against every body instead of being binned) */
static float body_bounding_radius(const sgo_body* b) { if (b->shape_type == SGP_SHAPE_MESH) return 3.0e38f; return b->shape_type == SGP_SHAPE_HULL ? b->hull->bound_radius : shape_bounding_radius(b->shape_type, b->shape); }

static void body_update_aabb(sgo_body* b)
{
	v3 e;
	if (b->shape_type == SGP_SHAPE_MESH) {
		/* corners of the mesh's local bounds */
		const m33 R = quat_to_m33(b->rot);
		v3 mn = V3(3.4e38f, 3.4e38f, 3.4e38f), mx = V3(-3.4e38f, -3.4e38f, -3.4e38f);
		for (int k = 0; k < 8; ++k) {
			const v3 c = V3((k & 1) ? b->mesh->aabb_max.x : b->mesh->aabb_min.x, (k & 2) ? b->mesh->aabb_max.y : b->mesh->aabb_min.y, (k & 4) ? b->mesh->aabb_max.z : b->mesh->aabb_min.z);
			const v3 p = m33_mul(R, c); mn = v3_min(mn, p); mx = v3_max(mx, p);
		}
		b->aabb_min = v3_add(b->pos, mn); b->aabb_max = v3_add(b->pos, mx);
		return;
	}
	if (b->shape_type == SGP_SHAPE_HULL) {
		const m33 R = quat_to_m33(b->rot);
		v3 mn = V3(3.4e38f, 3.4e38f, 3.4e38f), mx = V3(-3.4e38f, -3.4e38f, -3.4e38f);
		for (int i = 0; i < b->hull->nv; ++i) { const v3 p = m33_mul(R, b->hull->verts[i]); mn = v3_min(mn, p); mx = v3_max(mx, p); }
		b->aabb_min = v3_add(b->pos, mn); b->aabb_max = v3_add(b->pos, mx);
		return;
	}
	if (b->shape_type == SGP_SHAPE_SPHERE) e = V3(b->shape[0], b->shape[0], b->shape[0]);
	else {
		const m33 R = quat_to_m33(b->rot);
		if (b->shape_type == SGP_SHAPE_BOX) {
			const float hx = b->shape[0], hy = b->shape[1], hz = b->shape[2];
			e = V3(fabsf(R.c0.x) * hx + fabsf(R.c1.x) * hy + fabsf(R.c2.x) * hz,
			       fabsf(R.c0.y) * hx + fabsf(R.c1.y) * hy + fabsf(R.c2.y) * hz,
			       fabsf(R.c0.z) * hx + fabsf(R.c1.z) * hy + fabsf(R.c2.z) * hz);
		} else {
			const float r = b->shape[0], hh = b->shape[1];
			e = V3(fabsf(R.c2.x) * hh + r, fabsf(R.c2.y) * hh + r, fabsf(R.c2.z) * hh + r);
		}
	}
	b->aabb_min = v3_sub(b->pos, e);
	b->aabb_max = v3_add(b->pos, e);
}

/* Body::GetSleepTestPoints: COM plus two points on the two largest local extents. */
static void body_sleep_points(const sgo_body* b, v3 out[3])
{
	const v3 ext = shape_local_half_h(b->shape_type, b->shape, b->hull);
	const m33 R = quat_to_m33(b->rot);
	int lowest = 0;
	if (ext.y < v3_get(ext, lowest)) lowest = 1;
	if (ext.z < v3_get(ext, lowest)) lowest = 2;
	const int i1 = lowest == 0 ? 1 : 0;
	const int i2 = lowest == 2 ? 1 : 2;
	out[0] = b->pos;
	out[1] = v3_add(b->pos, v3_scale(m33_col(R, i1), v3_get(ext, i1)));
	out[2] = v3_add(b->pos, v3_scale(m33_col(R, i2), v3_get(ext, i2)));
}

static void body_reset_sleep(sgo_body* b)
{
	v3 p[3];
	body_sleep_points(b, p);
	for (int i = 0; i < 3; ++i) { b->sleep_c[i] = p[i]; b->sleep_r[i] = 0.0f; }
	b->sleep_timer = 0.0f;
}

/* ------------------------------------------------------------------------------------------------ */
/* events                                                                                           */

#define PUSH_EVENT(arr, n, cap, type, val) do { \
	if ((n) == (cap)) { (cap) = (cap) ? (cap) * 2 : 64; (arr) = (type*)realloc((arr), (size_t)(cap) * sizeof(type)); } \
	(arr)[(n)++] = (val); } while (0)

static void push_body_event(sgo_world* w, int kind, uint32_t id)
{
	sgp_body_event e; e.id = id; e._pad = 0; e.userdata = w->bodies[id].userdata;
	if (kind == SGP_EVENT_ACTIVATED) { PUSH_EVENT(w->ev_act, w->n_act, w->cap_act, sgp_body_event, e); w->tot_act++; }
	else if (kind == SGP_EVENT_DEACTIVATED) { PUSH_EVENT(w->ev_deact, w->n_deact, w->cap_deact, sgp_body_event, e); w->tot_deact++; }
	else PUSH_EVENT(w->ev_water, w->n_water, w->cap_water, sgp_body_event, e);
}

static void body_activate(sgo_world* w, uint32_t id)
{
	sgo_body* b = &w->bodies[id];
	if (!b->alive || b->motion == SGP_MOTION_STATIC) return;
	if (!b->active) {
		b->active = 1;
		push_body_event(w, SGP_EVENT_ACTIVATED, id);
	}
	body_reset_sleep(b);
}

/* ------------------------------------------------------------------------------------------------ */
/* lifecycle                                                                                        */

SGO_API int sgo_world_create(const sgp_world_desc* desc, sgo_world** out)
{
	if (!desc || !out || desc->max_bodies == 0) return SGP_ERR_INVALID;
	sgo_world* w = (sgo_world*)calloc(1, sizeof(sgo_world));
	w->desc = *desc;
	w->st = desc->settings;
	w->gravity = V3(desc->gravity[0], desc->gravity[1], desc->gravity[2]);
	w->cap = desc->max_bodies;
	w->bodies = (sgo_body*)calloc(w->cap, sizeof(sgo_body));
	w->free_list = (uint32_t*)malloc(sizeof(uint32_t) * w->cap);
	w->free_triples = (uint32_t*)malloc(sizeof(uint32_t) * (w->cap / 3 + 1)); w->n_free_triples = 0;
	w->free_mesh_ids = NULL; w->n_free_mesh_ids = 0; w->free_hull_ids = NULL; w->n_free_hull_ids = 0;
	w->cell_keys = (uint64_t*)malloc(sizeof(uint64_t) * w->cap);
	w->cell_idx = (uint32_t*)malloc(sizeof(uint32_t) * w->cap);
	w->large = (uint32_t*)malloc(sizeof(uint32_t) * w->cap);
	w->is_ghost = (int*)calloc(w->cap, sizeof(int));
	if (w->desc.large_body_radius <= 0.0f) w->desc.large_body_radius = 4.0f;
	w->cap_hulls = 16; w->hulls = (sgo_hull**)calloc(w->cap_hulls, sizeof(sgo_hull*));
	w->hulls[0] = (sgo_hull*)malloc(sizeof(sgo_hull)); sgo_hull_cube_template(w->hulls[0]); w->n_hulls = 1;
	w->cap_meshes = 16; w->meshes = (sgo_mesh**)calloc(w->cap_meshes, sizeof(sgo_mesh*)); w->n_meshes = 1;
	*out = w;
	return SGP_OK;
}

SGO_API int sgo_world_destroy(sgo_world* w)
{
	if (!w) return SGP_ERR_INVALID;
	free(w->bodies); free(w->free_list); free(w->free_triples); free(w->free_mesh_ids); free(w->free_hull_ids); free(w->pairs); free(w->cons); free(w->prev);
	free(w->prev_keys_sorted); free(w->prev_idx_sorted); free(w->order);
	free(w->ev_act); free(w->ev_deact); free(w->ev_water); free(w->ev_added); free(w->ev_pers);
	free(w->bp_ki); free(w->cell_keys); free(w->cell_idx); free(w->large); free(w->is_ghost); free(w->ghost_gid); free(w->ghost_lid);
	free(w->vehicles);
	for (uint32_t k = 0; k < w->n_hulls; ++k) free(w->hulls[k]);       /* (free(NULL) for destroyed hulls) */
	free(w->hulls);
	for (uint32_t k = 1; k < w->n_meshes; ++k) if (w->meshes[k]) { free(w->meshes[k]->verts); free(w->meshes[k]->tris); free(w->meshes[k]->mats); free(w->meshes[k]); }
	free(w->meshes);
	free(w);
	return SGP_OK;
}

static int finite3(const float* v) { return isfinite(v[0]) && isfinite(v[1]) && isfinite(v[2]); }

/* addObject, PhysicsWorld.cpp:1169-1311 */
SGO_API int sgo_body_add(sgo_world* w, const sgp_body_desc* d, uint32_t* id_out)
{
	if (!w || !d) return SGP_ERR_INVALID;
	if (!finite3(d->pos) || fabsf(d->pos[0]) > 1.0e9f || fabsf(d->pos[1]) > 1.0e9f || fabsf(d->pos[2]) > 1.0e9f) return SGP_ERR_REJECTED; /* :1178 */
	if (d->shape_type < 0 || d->shape_type > SGP_SHAPE_MESH) return SGP_ERR_INVALID;
	const sgo_hull* hull = NULL;
	const sgo_mesh* mesh = NULL;
	if (d->shape_type == SGP_SHAPE_MESH) {
		const uint32_t mid = (uint32_t)d->shape[0];
		if (!(d->shape[0] >= 1.0f) || (float)mid != d->shape[0] || mid >= w->n_meshes || !w->meshes[mid]) return SGP_ERR_INVALID;
		if (d->motion_type == SGP_MOTION_DYNAMIC) return SGP_ERR_INVALID;      /* JPH::MeshShape: static and kinematic bodies (the reference turns a dynamic mesh object into a kinematic one, PhysicsWorld.cpp:1290) */
		mesh = w->meshes[mid];
	}
	if (d->shape_type == SGP_SHAPE_HULL) {
		const uint32_t hid = (uint32_t)d->shape[0];
		if (!(d->shape[0] >= 1.0f) || (float)hid != d->shape[0] || hid >= w->n_hulls || !w->hulls[hid]) return SGP_ERR_INVALID;   /* hull 0 is the internal cube template */
		hull = w->hulls[hid];
	} else if (d->shape_type == SGP_SHAPE_BOX) hull = w->hulls[0];
	const int nparam = d->shape_type == SGP_SHAPE_BOX ? 3 : (d->shape_type == SGP_SHAPE_SPHERE ? 1 : ((d->shape_type == SGP_SHAPE_HULL || d->shape_type == SGP_SHAPE_MESH) ? 0 : 2));
	for (int i = 0; i < nparam; ++i) {
		const float lim = (d->shape_type == SGP_SHAPE_CAPSULE && i == 1) ? 0.0f : 0.5e-7f; /* |scale| < 1e-7 on a 0.5 unit shape, :1184 */
		if (!isfinite(d->shape[i]) || d->shape[i] < lim) return SGP_ERR_REJECTED;
	}
	uint32_t id;
	if (mesh) {
		/* three consecutive slots: the body and its two aliases -- a triple a removed mesh body left behind, else fresh ones */
		if (w->n_free_triples) id = w->free_triples[--w->n_free_triples];
		else { if (w->high + 3 > w->cap) return SGP_ERR_CAPACITY; id = w->high; w->high += 3; }
	}
	else if (w->n_free) id = w->free_list[--w->n_free];
	else { if (w->high >= w->cap) return SGP_ERR_CAPACITY; id = w->high++; }
	sgo_body* b = &w->bodies[id];
	{ const uint32_t gen_ = b->slot_gen; memset(b, 0, sizeof(*b)); b->slot_gen = gen_; }      /* (the slot's generation outlives its bodies) */
	b->pos = V3(d->pos[0], d->pos[1], d->pos[2]);
	quat q = { d->rot[0], d->rot[1], d->rot[2], d->rot[3] };
	b->rot = q;
	b->linv = V3(d->lin_vel[0], d->lin_vel[1], d->lin_vel[2]);
	b->angv = V3(d->ang_vel[0], d->ang_vel[1], d->ang_vel[2]);
	b->shape_type = d->shape_type;
	memcpy(b->shape, d->shape, sizeof(b->shape));
	b->hull = hull;
	b->mesh = mesh;
	b->motion = d->motion_type; b->layer = d->layer;
	b->friction = clampf(d->friction, 0.0f, 1.0f);         /* :1236 */
	b->restitution = clampf(d->restitution, 0.0f, 1.0f);   /* :1237 */
	b->mass = fmaxf(0.001f, d->mass);                      /* :1238 */
	b->gravity_factor = d->gravity_factor;
	b->lin_damp = d->linear_damping; b->ang_damp = d->angular_damping;
	b->is_sensor = d->is_sensor; b->allow_sleep = d->allow_sleeping; b->zero_lin_drag = d->use_zero_linear_drag;
	b->cache_invalid = 1;
	b->userdata = d->userdata;
	if (b->motion == SGP_MOTION_DYNAMIC) mass_properties(b->shape_type, b->shape, b->hull, b->mass, &b->inv_mass, &b->inv_inertia);
	else { b->inv_mass = 0.0f; b->inv_inertia = V3(0.0f, 0.0f, 0.0f); }
	if (b->motion != SGP_MOTION_DYNAMIC) { /* non-dynamic bodies carry no force */ }
	b->alive = 1; b->active = 0;
	/* a label names a slot AND the generation of the body in it: a body created in the slot a removed island root left must not share the wake label of that
	   island's sleepers (round 6) */
	b->slot_gen = (b->slot_gen + 1u) & 0x7Fu;
	b->sleep_label = SGO_LABEL(id, b->slot_gen);
	b->comp_root = SGP_INVALID_ID;
	body_update_aabb(b);
	body_reset_sleep(b);
	w->n_alive++;
	if (mesh) for (uint32_t k = 1; k <= 2; ++k) { const uint32_t ag = (w->bodies[id + k].slot_gen + 1u) & 0x7Fu; w->bodies[id + k] = *b; w->bodies[id + k].is_alias = 1; w->bodies[id + k].slot_gen = ag; w->bodies[id + k].sleep_label = SGO_LABEL(id + k, ag); }
	if (d->activate) body_activate(w, id);
	if (id_out) *id_out = id;
	return SGP_OK;
}

SGO_API int sgo_body_add_batch(sgo_world* w, const sgp_body_desc* d, uint32_t n, uint32_t* ids_out)
{
	for (uint32_t i = 0; i < n; ++i) {
		uint32_t id = SGP_INVALID_ID;
		const int r = sgo_body_add(w, &d[i], &id);
		if (r != SGP_OK && r != SGP_ERR_REJECTED) return r;
		if (ids_out) ids_out[i] = (r == SGP_OK) ? id : SGP_INVALID_ID;
	}
	return SGP_OK;
}

static int live(sgo_world* w, uint32_t id) { return w && id < w->high && w->bodies[id].alive; }

/* ---- static compound bodies (StaticCompoundShapeSettings, MeshBuilding.cpp:396-407): one body slot per child ---------------------- */
static void compound_child_pose(const float P[3], const float R[4], const float lp[3], const float lr[4], float pos_out[3], float rot_out[4])
{
	const float x = R[0], y = R[1], z = R[2], w_ = R[3];
	const float vx = lp[0], vy = lp[1], vz = lp[2];
	const float tx = 2.0f * (y * vz - z * vy), ty = 2.0f * (z * vx - x * vz), tz = 2.0f * (x * vy - y * vx);
	pos_out[0] = P[0] + (vx + w_ * tx + (y * tz - z * ty));
	pos_out[1] = P[1] + (vy + w_ * ty + (z * tx - x * tz));
	pos_out[2] = P[2] + (vz + w_ * tz + (x * ty - y * tx));
	const float ox = lr[0], oy = lr[1], oz = lr[2], ow = lr[3];
	rot_out[0] = w_ * ox + x * ow + y * oz - z * oy;
	rot_out[1] = w_ * oy - x * oz + y * ow + z * ox;
	rot_out[2] = w_ * oz + x * oy - y * ox + z * ow;
	rot_out[3] = w_ * ow - x * ox - y * oy - z * oz;
}
static int is_compound_child(const sgo_world* w, uint32_t id) { return w->bodies[id].comp_root != SGP_INVALID_ID && w->bodies[id].comp_root != id; }
static int is_compound(const sgo_world* w, uint32_t id) { return w->bodies[id].comp_root == id; }
/* the slots of compound `root`'s children, in child order (children are found by their back reference) */
static uint32_t compound_children(const sgo_world* w, uint32_t root, uint32_t* ids)
{
	const uint32_t n = w->bodies[root].comp_n;
	for (uint32_t i = 0; i < w->high; ++i) { const sgo_body* b = &w->bodies[i]; if (b->alive && !b->is_alias && b->comp_root == root && b->comp_child < n) ids[b->comp_child] = i; }
	return n;
}
static uint32_t compound_id_of(const sgo_world* w, uint32_t id, uint32_t* sub)
{
	const sgo_body* b = &w->bodies[id];
	if (b->comp_root == SGP_INVALID_ID) { if (sub) *sub = 0; return id; }
	if (sub) *sub = b->comp_child;
	return b->comp_root;
}
SGO_API int sgo_body_remove(sgo_world* w, uint32_t id);
SGO_API int sgo_body_add_compound(sgo_world* w, const sgp_body_desc* base, const sgp_compound_child* children, uint32_t n, uint32_t* id_out)
{
	if (!w || !base || !children || !id_out) return SGP_ERR_INVALID;
	*id_out = SGP_INVALID_ID;
	if (n < 1 || n > SGP_MAX_COMPOUND_CHILDREN) return SGP_ERR_INVALID;
	if (base->motion_type != SGP_MOTION_STATIC) return SGP_ERR_INVALID;
	if (!finite3(base->pos) || !isfinite(base->rot[0]) || !isfinite(base->rot[1]) || !isfinite(base->rot[2]) || !isfinite(base->rot[3])) return SGP_ERR_REJECTED;
	for (uint32_t k = 0; k < n; ++k) for (int a = 0; a < 4; ++a) if ((a < 3 && !isfinite(children[k].pos[a])) || !isfinite(children[k].rot[a])) return SGP_ERR_INVALID;
	uint32_t ids[SGP_MAX_COMPOUND_CHILDREN];
	for (uint32_t k = 0; k < n; ++k) {
		sgp_body_desc d = *base;
		d.shape_type = children[k].shape_type; memcpy(d.shape, children[k].shape, sizeof(d.shape));
		compound_child_pose(base->pos, base->rot, children[k].pos, children[k].rot, d.pos, d.rot);
		d.activate = 0;
		const int r = sgo_body_add(w, &d, &ids[k]);
		if (r != SGP_OK) { for (uint32_t j = 0; j < k; ++j) sgo_body_remove(w, ids[j]); return r; }
	}
	for (uint32_t k = 0; k < n; ++k) {
		sgo_body* b = &w->bodies[ids[k]];
		b->comp_root = ids[0]; b->comp_child = k;
		memcpy(b->comp_local_pos, children[k].pos, 12); memcpy(b->comp_local_rot, children[k].rot, 16);
		if (b->shape_type == SGP_SHAPE_MESH) for (int a = 1; a <= 2; ++a) { w->bodies[ids[k] + a].comp_root = SGP_INVALID_ID; }
	}
	sgo_body* r0 = &w->bodies[ids[0]];
	r0->comp_n = n; memcpy(r0->comp_pos, base->pos, 12); memcpy(r0->comp_rot, base->rot, 16);
	w->n_alive -= (n - 1);
	*id_out = ids[0];
	return SGP_OK;
}
SGO_API int sgo_body_compound_size(sgo_world* w, uint32_t id, uint32_t* n_out)
{
	if (!live(w, id) || !n_out) return SGP_ERR_BAD_ID;
	*n_out = is_compound(w, id) ? w->bodies[id].comp_n : 0u;
	return SGP_OK;
}
static int set_pose_one(sgo_world* w, uint32_t id, const float pos[3], const float rot[4]);
/* a pose edit of a compound: every child gets the compound's new pose composed with its own.  rot == NULL keeps the rotation. */
static void compound_set_pose(sgo_world* w, uint32_t root, const float pos[3], const float* rot)
{
	sgo_body* r0 = &w->bodies[root];
	memcpy(r0->comp_pos, pos, 12); if (rot) memcpy(r0->comp_rot, rot, 16);
	uint32_t ids[SGP_MAX_COMPOUND_CHILDREN];
	const uint32_t n = compound_children(w, root, ids);
	for (uint32_t k = 0; k < n; ++k) {
		float p[3], q[4];
		sgo_body* b = &w->bodies[ids[k]];
		compound_child_pose(r0->comp_pos, r0->comp_rot, b->comp_local_pos, b->comp_local_rot, p, q);
		set_pose_one(w, ids[k], p, q);
	}
}
/* a mesh body owns the two alias slots behind it (second / third contact manifold of a pair): they share its pose */
static void sync_mesh_aliases(sgo_world* w, uint32_t id)
{
	if (w->bodies[id].shape_type != SGP_SHAPE_MESH || w->bodies[id].is_alias) return;
	/* (a kinematic mesh body -- a scripted platform -- shares its velocities with them too; they are never awake themselves) */
	for (int k = 1; k <= 2; ++k) { sgo_body* a = &w->bodies[id + k]; a->pos = w->bodies[id].pos; a->rot = w->bodies[id].rot; a->linv = w->bodies[id].linv; a->angv = w->bodies[id].angv; a->active = 0; }
}

SGO_API int sgo_body_remove(sgo_world* w, uint32_t id)
{
	if (!live(w, id)) return SGP_ERR_BAD_ID;
	for (uint32_t k = 0; k < w->n_vehicles; ++k) if (w->vehicles[k].alive && w->vehicles[k].body == id) w->vehicles[k].alive = 0;   /* a vehicle does not outlive its chassis */
	if (w->bodies[id].is_alias) return SGP_ERR_BAD_ID;
	if (is_compound_child(w, id)) return SGP_ERR_BAD_ID;
	if (is_compound(w, id)) {
		uint32_t ids[SGP_MAX_COMPOUND_CHILDREN];
		const uint32_t n = compound_children(w, id, ids);
		for (uint32_t k = 0; k < n; ++k) w->bodies[ids[k]].comp_root = SGP_INVALID_ID;
		for (uint32_t k = 1; k < n; ++k) { const int r = sgo_body_remove(w, ids[k]); if (r != SGP_OK) return r; w->n_alive++; }
	}
	const int nslots = w->bodies[id].shape_type == SGP_SHAPE_MESH ? 3 : 1;
	for (int k = 0; k < nslots; ++k) { w->bodies[id + k].alive = 0; w->bodies[id + k].active = 0; w->bodies[id + k].is_alias = 0; }
	if (nslots == 3) w->free_triples[w->n_free_triples++] = id; else w->free_list[w->n_free++] = id;
	w->n_alive--;
	return SGP_OK;
}
SGO_API int sgo_body_activate(sgo_world* w, uint32_t id) { if (!live(w, id)) return SGP_ERR_BAD_ID; body_activate(w, id); return SGP_OK; }
SGO_API int sgo_body_get_volume(sgo_world* w, uint32_t id, float* out) { if (!live(w, id) || !out) return SGP_ERR_BAD_ID; *out = shape_volume_h(w->bodies[id].shape_type, w->bodies[id].shape, w->bodies[id].hull); return SGP_OK; }
SGO_API int sgo_body_set_layer(sgo_world* w, uint32_t id, int32_t layer)
{
	if (!live(w, id) || is_compound_child(w, id)) return SGP_ERR_BAD_ID;
	if (is_compound(w, id)) { uint32_t ids[SGP_MAX_COMPOUND_CHILDREN]; const uint32_t n = compound_children(w, id, ids); for (uint32_t k = 0; k < n; ++k) w->bodies[ids[k]].layer = layer; return SGP_OK; }
	w->bodies[id].layer = layer;
	return SGP_OK;
}

static int set_pose_one(sgo_world* w, uint32_t id, const float pos[3], const float rot[4])
{
	sgo_body* b = &w->bodies[id];
	b->pos = V3(pos[0], pos[1], pos[2]);
	quat q = { rot[0], rot[1], rot[2], rot[3] }; b->rot = q;
	body_update_aabb(b);
	sync_mesh_aliases(w, id);
	return SGP_OK;
}
SGO_API int sgo_body_set_pose_vel(sgo_world* w, uint32_t id, const float pos[3], const float rot[4], const float lv[3], const float av[3])
{
	if (!live(w, id) || is_compound_child(w, id)) return SGP_ERR_BAD_ID;
	if (is_compound(w, id)) { compound_set_pose(w, id, pos, rot); return SGP_OK; }
	sgo_body* b = &w->bodies[id];
	b->pos = V3(pos[0], pos[1], pos[2]);
	quat q = { rot[0], rot[1], rot[2], rot[3] }; b->rot = q;
	if (b->motion != SGP_MOTION_STATIC) { b->linv = V3(lv[0], lv[1], lv[2]); b->angv = V3(av[0], av[1], av[2]); }
	body_update_aabb(b);
	sync_mesh_aliases(w, id);
	return SGP_OK;
}
SGO_API int sgo_body_set_pose_shape(sgo_world* w, uint32_t id, const float pos[3], const float rot[4], const float shape[4])
{
	if (!live(w, id) || is_compound_child(w, id)) return SGP_ERR_BAD_ID;
	if (is_compound(w, id)) { compound_set_pose(w, id, pos, rot); return SGP_OK; }
	sgo_body* b = &w->bodies[id];
	b->pos = V3(pos[0], pos[1], pos[2]);
	quat q = { rot[0], rot[1], rot[2], rot[3] }; b->rot = q;
	b->linv = V3(0, 0, 0); b->angv = V3(0, 0, 0);
	if (b->shape_type != SGP_SHAPE_HULL && b->shape_type != SGP_SHAPE_MESH) { memcpy(b->shape, shape, sizeof(b->shape)); b->cache_invalid = 1; }   /* inUpdateMassProperties = false, PhysicsWorld.cpp:579; hulls and meshes are pre-scaled */
	body_update_aabb(b);
	sync_mesh_aliases(w, id);
	body_activate(w, id);
	return SGP_OK;
}
SGO_API int sgo_body_set_pos(sgo_world* w, uint32_t id, const float pos[3])
{
	if (!live(w, id) || is_compound_child(w, id)) return SGP_ERR_BAD_ID;
	if (is_compound(w, id)) { compound_set_pose(w, id, pos, NULL); return SGP_OK; }
	w->bodies[id].pos = V3(pos[0], pos[1], pos[2]);
	body_update_aabb(&w->bodies[id]);
	sync_mesh_aliases(w, id);
	return SGP_OK;
}
SGO_API int sgo_body_set_vel(sgo_world* w, uint32_t id, const float lv[3], const float av[3])
{
	if (!live(w, id)) return SGP_ERR_BAD_ID;
	if (w->bodies[id].motion == SGP_MOTION_STATIC) return SGP_OK;
	w->bodies[id].linv = V3(lv[0], lv[1], lv[2]); w->bodies[id].angv = V3(av[0], av[1], av[2]);
	sync_mesh_aliases(w, id);
	return SGP_OK;
}
/* MotionProperties::MoveKinematic: velocities that reach the target in dt. Host-side maths (acos), like the product. */
SGO_API int sgo_body_move_kinematic(sgo_world* w, uint32_t id, const float tp[3], const float tr[4], float dt)
{
	if (!live(w, id)) return SGP_ERR_BAD_ID;
	sgo_body* b = &w->bodies[id];
	if (b->motion != SGP_MOTION_KINEMATIC || dt <= 0.0f) return SGP_OK;
	b->linv = v3_scale(v3_sub(V3(tp[0], tp[1], tp[2]), b->pos), 1.0f / dt);
	quat t = { tr[0], tr[1], tr[2], tr[3] };
	quat c = { -b->rot.x, -b->rot.y, -b->rot.z, b->rot.w };
	quat dq = quat_mul(t, c);
	if (dq.w < 0.0f) { dq.x = -dq.x; dq.y = -dq.y; dq.z = -dq.z; dq.w = -dq.w; }
	const float sl = sqrtf(dq.x * dq.x + dq.y * dq.y + dq.z * dq.z);
	if (sl > 1.0e-12f) {
		const float angle = sgo_quat_angle(sl, dq.w);
		b->angv = v3_scale(V3(dq.x / sl, dq.y / sl, dq.z / sl), angle / dt);
	} else b->angv = V3(0, 0, 0);
	body_activate(w, id);
	sync_mesh_aliases(w, id);
	return SGP_OK;
}
SGO_API int sgo_body_add_force(sgo_world* w, uint32_t id, const float f[3])
{
	if (!live(w, id)) return SGP_ERR_BAD_ID;
	sgo_body* b = &w->bodies[id];
	if (b->motion != SGP_MOTION_DYNAMIC) return SGP_OK;
	b->force = v3_add(b->force, V3(f[0], f[1], f[2]));
	body_activate(w, id);
	return SGP_OK;
}
SGO_API int sgo_body_add_torque(sgo_world* w, uint32_t id, const float t[3])
{
	if (!live(w, id)) return SGP_ERR_BAD_ID;
	sgo_body* b = &w->bodies[id];
	if (b->motion != SGP_MOTION_DYNAMIC) return SGP_OK;
	b->torque = v3_add(b->torque, V3(t[0], t[1], t[2]));
	body_activate(w, id);
	return SGP_OK;
}
SGO_API int sgo_body_add_force_at(sgo_world* w, uint32_t id, const float f[3], const float p[3])
{
	if (!live(w, id)) return SGP_ERR_BAD_ID;
	sgo_body* b = &w->bodies[id];
	if (b->motion != SGP_MOTION_DYNAMIC) return SGP_OK;
	const v3 F = V3(f[0], f[1], f[2]);
	b->force = v3_add(b->force, F);
	b->torque = v3_add(b->torque, v3_cross(v3_sub(V3(p[0], p[1], p[2]), b->pos), F));
	body_activate(w, id);
	return SGP_OK;
}

static void fill_state(const sgo_body* b, uint32_t id, sgp_body_state* s)
{
	s->pos[0] = b->pos.x; s->pos[1] = b->pos.y; s->pos[2] = b->pos.z;
	s->rot[0] = b->rot.x; s->rot[1] = b->rot.y; s->rot[2] = b->rot.z; s->rot[3] = b->rot.w;
	s->lin_vel[0] = b->linv.x; s->lin_vel[1] = b->linv.y; s->lin_vel[2] = b->linv.z;
	s->ang_vel[0] = b->angv.x; s->ang_vel[1] = b->angv.y; s->ang_vel[2] = b->angv.z;
	s->active = (uint32_t)b->active; s->underwater = (uint32_t)b->underwater; s->submerged_volume = b->submerged;
	s->id = b->alive ? id : SGP_INVALID_ID;
}
SGO_API int sgo_body_get_state(sgo_world* w, const uint32_t* ids, uint32_t n, sgp_body_state* out)
{
	for (uint32_t i = 0; i < n; ++i) { if (!live(w, ids[i])) return SGP_ERR_BAD_ID; fill_state(&w->bodies[ids[i]], ids[i], &out[i]); }
	return SGP_OK;
}
SGO_API int sgo_world_read_states(sgo_world* w, uint32_t first, uint32_t n, sgp_body_state* out)
{
	if (!w || first + n > w->cap) return SGP_ERR_INVALID;
	for (uint32_t i = 0; i < n; ++i) fill_state(&w->bodies[first + i], first + i, &out[i]);
	return SGP_OK;
}
SGO_API int sgo_world_read_active(sgo_world* w, sgp_body_state* out, uint32_t cap, uint32_t* n_out)
{
	uint32_t n = 0;
	for (uint32_t i = 0; i < w->high; ++i) if (w->bodies[i].alive && w->bodies[i].active) { if (n < cap) fill_state(&w->bodies[i], i, &out[n]); ++n; }
	*n_out = n;
	return SGP_OK;
}
SGO_API int sgo_world_set_water(sgo_world* w, int enabled, float z) { w->water_enabled = enabled; w->water_z = z; return SGP_OK; }
SGO_API int sgo_world_set_contact_events(sgo_world* w, int enabled) { w->contact_events = enabled; return SGP_OK; }
SGO_API int sgo_world_num_bodies(sgo_world* w, uint32_t* n) { *n = w->n_alive; return SGP_OK; }

/* ------------------------------------------------------------------------------------------------ */
/* broad phase: uniform grid over the small bodies (cell >= largest small-body AABB + margin), large  */
/* bodies (ground quad, PhysicsWorld.cpp:1123) against everything.  Only the resulting SET matters.   */

static int body_is_active_for_pairs(const sgo_body* b)
{
	if (!b->alive) return 0;
	if (b->motion == SGP_MOTION_DYNAMIC) return b->active;
	if (b->motion == SGP_MOTION_KINEMATIC) return b->active;
	return 0;
}

/* everything of the pair test but the "one of them is active" condition */
static int pair_passes_geom(const sgo_world* w, uint32_t i, uint32_t j)
{
	const sgo_body* a = &w->bodies[i]; const sgo_body* b = &w->bodies[j];
	if (a->motion != SGP_MOTION_DYNAMIC && b->motion != SGP_MOTION_DYNAMIC) return 0;
	if (!(layers_collide(a->layer, b->layer))) return 0;
	const float d = w->st.speculative_contact_distance;
	if (a->aabb_min.x - d > b->aabb_max.x || b->aabb_min.x - d > a->aabb_max.x) return 0;
	if (a->aabb_min.y - d > b->aabb_max.y || b->aabb_min.y - d > a->aabb_max.y) return 0;
	if (a->aabb_min.z - d > b->aabb_max.z || b->aabb_min.z - d > a->aabb_max.z) return 0;
	return 1;
}

static int pair_passes(const sgo_world* w, uint32_t i, uint32_t j)
{
	if (!(body_is_active_for_pairs(&w->bodies[i]) || body_is_active_for_pairs(&w->bodies[j]))) return 0;
	return pair_passes_geom(w, i, j);
}

static void push_pair(sgo_world* w, uint32_t i, uint32_t j)
{
	if (w->n_pairs == w->cap_pairs) { w->cap_pairs = w->cap_pairs ? w->cap_pairs * 2 : 1024; w->pairs = (sgo_pair*)realloc(w->pairs, sizeof(sgo_pair) * w->cap_pairs); }
	sgo_pair p; p.a = i < j ? i : j; p.b = i < j ? j : i;
	w->pairs[w->n_pairs++] = p;
}

typedef struct keyidx_s { uint64_t key; uint32_t idx; } keyidx;
static int cmp_keyidx(const void* a, const void* b)
{
	const keyidx* x = (const keyidx*)a; const keyidx* y = (const keyidx*)b;
	if (x->key < y->key) return -1; if (x->key > y->key) return 1;
	return (x->idx > y->idx) - (x->idx < y->idx);
}

static uint64_t cell_key(int64_t cx, int64_t cy, int64_t cz)
{
	return ((uint64_t)(cz + (1 << 20)) << 42) | ((uint64_t)(cy + (1 << 20)) << 21) | (uint64_t)(cx + (1 << 20));
}

static uint32_t lower_bound_key(const keyidx* a, uint32_t n, uint64_t key)
{
	uint32_t lo = 0, hi = n;
	while (lo < hi) { const uint32_t mid = (lo + hi) / 2; if (a[mid].key < key) lo = mid + 1; else hi = mid; }
	return lo;
}

static void broad_phase(sgo_world* w)
{
	w->n_pairs = 0; w->n_large = 0;
	const float large_r = w->desc.large_body_radius;
	float cell = 0.5f;
	uint32_t n_small = 0;
	free(w->bp_ki);
	keyidx* ki = (keyidx*)malloc(sizeof(keyidx) * (w->high ? w->high : 1));
	w->bp_ki = ki;
	for (uint32_t i = 0; i < w->high; ++i) {
		const sgo_body* b = &w->bodies[i];
		if (!b->alive || b->is_alias) continue;
		if (body_bounding_radius(b) > large_r) { w->large[w->n_large++] = i; continue; }
		const v3 e = v3_sub(b->aabb_max, b->aabb_min);
		cell = fmaxf(cell, fmaxf(e.x, fmaxf(e.y, e.z)));
		ki[n_small++].idx = i;
	}
	cell += 2.0f * w->st.speculative_contact_distance;
	for (uint32_t k = 0; k < n_small; ++k) {
		const sgo_body* b = &w->bodies[ki[k].idx];
		const v3 c = v3_scale(v3_add(b->aabb_min, b->aabb_max), 0.5f);
		ki[k].key = cell_key((int64_t)floorf(c.x / cell), (int64_t)floorf(c.y / cell), (int64_t)floorf(c.z / cell));
	}
	qsort(ki, n_small, sizeof(keyidx), cmp_keyidx);
	w->bp_n_small = n_small; w->bp_cell = cell;
	#pragma omp parallel if (g_threads > 1)
	{
		sgo_pair* loc = NULL; uint32_t nloc = 0, caploc = 0;
		#pragma omp for schedule(dynamic, 256) nowait
		for (uint32_t k = 0; k < n_small; ++k) {
			const uint32_t i = ki[k].idx;
			if (!body_is_active_for_pairs(&w->bodies[i])) continue;
			const int64_t cx = (int64_t)(ki[k].key & 0x1FFFFF) - (1 << 20);
			const int64_t cy = (int64_t)((ki[k].key >> 21) & 0x1FFFFF) - (1 << 20);
			const int64_t cz = (int64_t)((ki[k].key >> 42) & 0x1FFFFF) - (1 << 20);
			for (int64_t dz = -1; dz <= 1; ++dz) for (int64_t dy = -1; dy <= 1; ++dy) for (int64_t dx = -1; dx <= 1; ++dx) {
				const uint64_t key = cell_key(cx + dx, cy + dy, cz + dz);
				for (uint32_t p = lower_bound_key(ki, n_small, key); p < n_small && ki[p].key == key; ++p) {
					const uint32_t j = ki[p].idx;
					if (j == i) continue;
					/* emitted once: by the lower id if both scan, else by the scanning (active) one */
					if (body_is_active_for_pairs(&w->bodies[j]) && j < i) continue;
					if (pair_passes(w, i, j)) {
						if (nloc == caploc) { caploc = caploc ? caploc * 2 : 1024; loc = (sgo_pair*)realloc(loc, sizeof(sgo_pair) * caploc); }
						loc[nloc].a = i < j ? i : j; loc[nloc].b = i < j ? j : i; ++nloc;
					}
				}
			}
		}
		#pragma omp critical
		{ for (uint32_t q = 0; q < nloc; ++q) push_pair(w, loc[q].a, loc[q].b); }
		free(loc);
	}
	for (uint32_t l = 0; l < w->n_large; ++l) {
		const uint32_t i = w->large[l];
		for (uint32_t j = 0; j < w->high; ++j) {
			if (j == i || !w->bodies[j].alive || w->bodies[j].is_alias) continue;
			/* large-large pairs once */
			if (body_bounding_radius(&w->bodies[j]) > large_r && j < i) continue;
			if (pair_passes(w, i, j)) push_pair(w, i, j);
		}
	}
}

/* ------------------------------------------------------------------------------------------------ */
/* contact constraints (Jolt ContactConstraintManager)                                               */

static sgo_shape body_shape_xf(const sgo_body* b)
{
	sgo_shape s; s.pos = b->pos; s.R = quat_to_m33(b->rot); s.type = b->shape_type;
	memcpy(s.p, b->shape, sizeof(s.p));
	s.hull = b->hull;
	return s;
}

static int body_movable(const sgo_body* b) { return b->motion == SGP_MOTION_DYNAMIC && b->active; }

/* AxisConstraintPart::CalculateConstraintProperties: 1 / (J M^-1 J^T) */
static float axis_eff_mass(float im1, sym33 I1, v3 r1, float im2, sym33 I2, v3 r2, v3 axis)
{
	float inv = 0.0f;
	if (im1 > 0.0f) { const v3 c = v3_cross(r1, axis); inv = im1 + v3_dot(sym33_mul(I1, c), c); }
	if (im2 > 0.0f) { const v3 c = v3_cross(r2, axis); inv = inv + (im2 + v3_dot(sym33_mul(I2, c), c)); }
	return inv > 0.0f ? 1.0f / inv : 0.0f;
}

static const sgo_constraint* find_prev(const sgo_world* w, uint64_t key)
{
	uint32_t lo = 0, hi = w->n_prev;
	while (lo < hi) { const uint32_t mid = (lo + hi) / 2; if (w->prev_keys_sorted[mid] < key) lo = mid + 1; else hi = mid; }
	if (lo < w->n_prev && w->prev_keys_sorted[lo] == key) return &w->prev[w->prev_idx_sorted[lo]];
	return NULL;
}

static void emit_contact_event(sgo_world* w, const sgo_constraint* c, const sgo_manifold* m, int persisted)
{
	sgp_contact_event e; memset(&e, 0, sizeof(e));
	const sgo_body* A = &w->bodies[c->a]; const sgo_body* B = &w->bodies[c->b];
	e.id1 = c->a; e.id2 = c->b; e.userdata1 = A->userdata; e.userdata2 = B->userdata;
	while (e.id1 > 0 && w->bodies[e.id1].is_alias) --e.id1;      /* a mesh body's alias slots report as the mesh body */
	while (e.id2 > 0 && w->bodies[e.id2].is_alias) --e.id2;
	e.id1 = compound_id_of(w, e.id1, NULL); e.id2 = compound_id_of(w, e.id2, NULL);      /* a compound's children report as the compound */
	e.lin_vel1[0] = A->linv.x; e.lin_vel1[1] = A->linv.y; e.lin_vel1[2] = A->linv.z;
	e.lin_vel2[0] = B->linv.x; e.lin_vel2[1] = B->linv.y; e.lin_vel2[2] = B->linv.z;
	e.base_offset[0] = m->p1[0].x; e.base_offset[1] = m->p1[0].y; e.base_offset[2] = m->p1[0].z;
	e.normal[0] = m->n.x; e.normal[1] = m->n.y; e.normal[2] = m->n.z;
	e.num_points = (uint32_t)m->np;
	float pen = -3.4e38f;
	for (int i = 0; i < m->np; ++i) {
		const v3 r = v3_sub(m->p1[i], m->p1[0]);
		e.rel_points_on1[i][0] = r.x; e.rel_points_on1[i][1] = r.y; e.rel_points_on1[i][2] = r.z;
		pen = fmaxf(pen, v3_dot(v3_sub(m->p1[i], m->p2[i]), m->n));
	}
	e.penetration = pen;
	if (persisted) PUSH_EVENT(w->ev_pers, w->n_pers, w->cap_pers, sgp_contact_event, e);
	else PUSH_EVENT(w->ev_added, w->n_added, w->cap_added, sgp_contact_event, e);
}

/* pose of body 2 relative to body 1: centre of mass offset in body 1's frame, conj(q1) * q2 */
static void pair_relative_pose(const sgo_body* A, const sgo_body* B, v3* dpos, quat* drot)
{
	*dpos = m33_tmul(quat_to_m33(A->rot), v3_sub(B->pos, A->pos));
	const quat ca = { -A->rot.x, -A->rot.y, -A->rot.z, A->rot.w };
	*drot = quat_mul(ca, B->rot);
}

/* ContactConstraintManager::GetContactsFromCache for one pair of non-mesh bodies: 1 = *m is last step's manifold carried to the bodies'
   current poses.  (Jolt also caches pairs WITHOUT contacts and skips their collision test; here only pairs that had a manifold are cached.
   Sensor pairs are always re-tested: their cached entry holds no points.)   UNVERIFIED: upstream */
static int reuse_cached_manifold(const sgo_world* w, uint32_t a, uint32_t b, sgo_manifold* m)
{
	if (!w->st.use_body_pair_contact_cache) return 0;
	const sgo_body* A = &w->bodies[a]; const sgo_body* B = &w->bodies[b];
	if (A->cache_invalid || B->cache_invalid || A->is_sensor || B->is_sensor) return 0;
	/* polytope pairs only (box / hull against box / hull): a deviation from Jolt, which caches every pair -- a sphere or capsule contact is
	   recomputed every step (cheap, and the same answer); see DESIGN.md */
	if (!((A->shape_type == SGP_SHAPE_BOX || A->shape_type == SGP_SHAPE_HULL) && (B->shape_type == SGP_SHAPE_BOX || B->shape_type == SGP_SHAPE_HULL))) return 0;
	const sgo_constraint* pc = find_prev(w, ((uint64_t)a << 32) | b);
	if (!pc) return 0;
	v3 dpos; quat drot;
	pair_relative_pose(A, B, &dpos, &drot);
	if (!(v3_len_sq(v3_sub(dpos, pc->dpos)) <= w->st.body_pair_cache_max_delta_position_sq)) return 0;
	const float dq = drot.x * pc->drot.x + drot.y * pc->drot.y + drot.z * pc->drot.z + drot.w * pc->drot.w;
	if (!(fabsf(dq) >= w->st.body_pair_cache_cos_max_delta_rotation_div2)) return 0;
	const m33 RA = quat_to_m33(A->rot), RB = quat_to_m33(B->rot);
	m->np = pc->np;
	m->n = m33_mul(RB, pc->nloc2);
	for (int i = 0; i < pc->np; ++i) { m->p1[i] = v3_add(A->pos, m33_mul(RA, pc->pt[i].local1)); m->p2[i] = v3_add(B->pos, m33_mul(RB, pc->pt[i].local2)); }
	return 1;
}

/* Constraint properties of manifold k (TemplatedAddContactConstraint); returns 0 for sensor pairs. Same arithmetic as the loop in find_contacts. */
static int setup_constraint(sgo_world* w, uint32_t k, const sgo_manifold* m, float dt, uint32_t* npts)
{
	sgo_constraint c = w->cons[k];
	const sgo_body* A = &w->bodies[c.a]; const sgo_body* B = &w->bodies[c.b];
	const sgo_constraint* pc = find_prev(w, c.key);
	c.persisted = pc != NULL;
	/* a sensor pair (mIsSensor, PhysicsWorld.cpp:1235) stays in the contact list with zero points: no response, but the pair is
	   cached so that the next step reports OnContactPersisted instead of OnContactAdded */
	if (A->is_sensor || B->is_sensor) c.np = 0;
	const m33 RA = quat_to_m33(A->rot), RB = quat_to_m33(B->rot);
	const float im1 = body_movable(A) ? A->inv_mass : 0.0f, im2 = body_movable(B) ? B->inv_mass : 0.0f;
	sym33 I1, I2; memset(&I1, 0, sizeof(I1)); memset(&I2, 0, sizeof(I2));
	if (im1 > 0.0f) I1 = world_inv_inertia(RA, A->inv_inertia);
	if (im2 > 0.0f) I2 = world_inv_inertia(RB, B->inv_inertia);
	c.friction = sqrtf(A->friction * B->friction);                      /* ContactConstraintManager's default combine: geometric mean   UNVERIFIED: upstream */
	const float restitution = fmaxf(A->restitution, B->restitution);    /* ... and the larger restitution                                UNVERIFIED: upstream */
	c.t1 = v3_normalized_perpendicular(c.n);
	c.t2 = v3_cross(c.n, c.t1);
	if (c.reused && pc) { c.dpos = pc->dpos; c.drot = pc->drot; c.nloc2 = pc->nloc2; }
	else { c.reused = 0; pair_relative_pose(A, B, &c.dpos, &c.drot); c.nloc2 = m33_tmul(RB, c.n); }
	for (int i = 0; i < c.np; ++i) {
		sgo_point* pt = &c.pt[i];
		const v3 p1 = m->p1[i], p2 = m->p2[i];
		pt->local1 = m33_tmul(RA, v3_sub(p1, A->pos));
		pt->local2 = m33_tmul(RB, v3_sub(p2, B->pos));
		if (c.reused) { pt->local1 = pc->pt[i].local1; pt->local2 = pc->pt[i].local2; }      /* the cached body-space points themselves: no drift from re-deriving them */
		pt->lam_n = pt->lam_t1 = pt->lam_t2 = 0.0f;
		if (pc && w->st.warm_start) {
			for (int j = 0; j < pc->np; ++j) {
				if (v3_len_sq(v3_sub(pt->local1, pc->pt[j].local1)) < w->st.contact_point_preserve_lambda_max_dist_sq &&
				    v3_len_sq(v3_sub(pt->local2, pc->pt[j].local2)) < w->st.contact_point_preserve_lambda_max_dist_sq) {
					pt->lam_n = pc->pt[j].lam_n; pt->lam_t1 = pc->pt[j].lam_t1; pt->lam_t2 = pc->pt[j].lam_t2;
					break;
				}
			}
		}
		const v3 mid = v3_scale(v3_add(p1, p2), 0.5f);
		pt->r1 = v3_sub(mid, A->pos); pt->r2 = v3_sub(mid, B->pos);
		const v3 va = v3_add(A->linv, v3_cross(A->angv, pt->r1));
		const v3 vb = v3_add(B->linv, v3_cross(B->angv, pt->r2));
		const float normal_velocity = v3_dot(v3_sub(vb, va), c.n);
		const float penetration = v3_dot(v3_sub(p1, p2), c.n);
		const float spec_bias = fmaxf(0.0f, -penetration / dt);
		float bias = spec_bias;
		if (restitution > 0.0f && normal_velocity < -w->st.min_velocity_for_restitution) {
			if (normal_velocity < -spec_bias) {
				v3 rel_acc = V3(0, 0, 0);
				if (im2 > 0.0f) rel_acc = v3_add(rel_acc, v3_scale(w->gravity, B->gravity_factor));
				if (im1 > 0.0f) rel_acc = v3_sub(rel_acc, v3_scale(w->gravity, A->gravity_factor));
				const float force_dv = fminf(0.0f, v3_dot(rel_acc, c.n)) * dt;
				bias = restitution * (normal_velocity - force_dv);
			}
		}
		pt->bias = bias;
		pt->eff_n = axis_eff_mass(im1, I1, pt->r1, im2, I2, pt->r2, c.n);
		pt->eff_t1 = axis_eff_mass(im1, I1, pt->r1, im2, I2, pt->r2, c.t1);
		pt->eff_t2 = axis_eff_mass(im1, I1, pt->r1, im2, I2, pt->r2, c.t2);
	}
	*npts = (uint32_t)c.np;
	w->cons[k] = c;
	return 1;
}

/* Narrow phase + constraint setup for every candidate pair. */
static int collide_with_mesh(const sgo_body* M, const sgo_shape* X, v3 lo, v3 hi, float max_sep, sgo_manifold* out, int active_edges, v3 movement);
static int g_active_edges = 1;      /* test switch: 0 = every edge collides with its own normal (what rounds 1-3 did), so that a KAT can show the difference */
SGO_API int sgo_set_active_edges(int on) { const int old = g_active_edges; g_active_edges = on ? 1 : 0; return old; }
SGO_API int sgo_mesh_edge_flags(sgo_world* w, uint32_t mesh_id, uint8_t* out, uint32_t cap);

/* manifolds of pairs [first, first + n): mans[3 (p - first) + g], hit[p - first] = number of manifolds, reused[p - first] */
static void collide_pairs(sgo_world* w, uint32_t first, uint32_t n, float dt, sgo_manifold* mans, unsigned char* hit, unsigned char* reused)
{
	#pragma omp parallel for schedule(static, 256) if (g_threads > 1)
	for (uint32_t q = 0; q < n; ++q) {
		const uint32_t p = first + q;
		const uint32_t a = w->pairs[p].a, b = w->pairs[p].b;
		const sgo_body* A = &w->bodies[a]; const sgo_body* B = &w->bodies[b];
		reused[q] = 0;
		if (A->shape_type == SGP_SHAPE_MESH || B->shape_type == SGP_SHAPE_MESH) {
			hit[q] = 0;
			if (A->shape_type == SGP_SHAPE_MESH && B->shape_type == SGP_SHAPE_MESH) continue;       /* (both static anyway) */
			const sgo_body* M = A->shape_type == SGP_SHAPE_MESH ? A : B; const sgo_body* X = M == A ? B : A;
			const sgo_shape sx = body_shape_xf(X);
			/* movement hint of the active-edge rule: X's velocity with this step's gravity, relative to the mesh (PhysicsSystem::ProcessBodyPair:
			   mActiveEdgeMovementDirection = v1 - v2, after ApplyGravity; the device has not applied the forces yet at this point and adds g dt itself) */
			const v3 mv = v3_sub(v3_add(X->linv_pre, v3_scale(v3_scale(w->gravity, X->gravity_factor), dt)), M->motion == SGP_MOTION_STATIC ? V3(0, 0, 0) : M->linv);
			hit[q] = (unsigned char)collide_with_mesh(M, &sx, X->aabb_min, X->aabb_max, w->st.speculative_contact_distance, &mans[3 * q], g_active_edges, mv);
			continue;
		}
		/* the body-pair contact cache: both bodies where they were (relative to each other) when the cached manifold was computed ->
		   the manifold is rebuilt from its body-space points instead of running the collision test */
		if (reuse_cached_manifold(w, a, b, &mans[3 * q])) { hit[q] = 1; reused[q] = 1; continue; }
		const sgo_shape sa = body_shape_xf(A), sb = body_shape_xf(B);
		hit[q] = (unsigned char)sgo_collide(&sa, &sb, w->st.speculative_contact_distance, &mans[3 * q]);
	}
}

/* ... appended to the constraint list (from slot *nm on) and, compacted, to out[] */
static void append_manifolds(sgo_world* w, uint32_t first, uint32_t n, const sgo_manifold* mans, const unsigned char* hit, const unsigned char* reused, sgo_manifold* out, uint32_t* nm_io)
{
	uint32_t nm = *nm_io;
	for (uint32_t q = 0; q < n; ++q) {
		const uint32_t p = first + q;
		for (int g = 0; g < hit[q]; ++g) {
			uint32_t a = w->pairs[p].a, b = w->pairs[p].b;
			sgo_manifold m = mans[3 * q + g];
			if (!(v3_len_sq(m.n) > 0.25f)) continue;          /* safety net: a manifold without a direction (or with a NaN one) is dropped, never solved */
			const int mesh_a = w->bodies[a].shape_type == SGP_SHAPE_MESH, mesh_b = w->bodies[b].shape_type == SGP_SHAPE_MESH;
			if (mesh_a || mesh_b) {
				/* the manifold runs mesh -> body; the constraint runs lower id -> higher id, with the mesh's g-th slot */
				const uint32_t mid = (mesh_a ? a : b) + (uint32_t)g, xid = mesh_a ? b : a;
				if (mid < xid) { a = mid; b = xid; } else { a = xid; b = mid; sgo_flip_manifold(&m); }
			}
			sgo_constraint* c = &w->cons[nm];
			memset(c, 0, sizeof(*c));
			c->a = a; c->b = b; c->key = ((uint64_t)a << 32) | b; c->prio = sgp_mix64(c->key);
			c->np = m.np; c->n = m.n; c->reused = reused[q];
			out[nm++] = m;
		}
	}
	*nm_io = nm;
}

static void cons_reserve(sgo_world* w, uint32_t n)
{
	if (w->cap_cons < n) { w->cap_cons = n + 1024; w->cons = (sgo_constraint*)realloc(w->cons, sizeof(sgo_constraint) * w->cap_cons); }
}

static int g_in_step_activation = 1;      /* test switch: 0 = a woken body meets its other contacts one step later (what rounds 1-3 did) */
SGO_API int sgo_set_in_step_activation(int on) { const int old = g_in_step_activation; g_in_step_activation = on ? 1 : 0; return old; }

static void find_contacts(sgo_world* w, float dt)
{
	w->n_cons = 0;
	/* pass 1: manifolds + activation of sleeping bodies touched by an active one.  A pair with a mesh body yields up to three
	   manifolds (groups of triangle contacts with similar normals), carried by the mesh body and its two alias slots. */
	const uint32_t n0 = w->n_pairs;
	cons_reserve(w, 3 * n0 + 1);
	sgo_manifold* mans = (sgo_manifold*)malloc(sizeof(sgo_manifold) * (n0 ? 3 * n0 : 1));
	uint32_t nm = 0;
	{
		sgo_manifold* raw = (sgo_manifold*)malloc(sizeof(sgo_manifold) * (n0 ? 3 * n0 : 1));
		unsigned char* hit = (unsigned char*)malloc(n0 ? n0 : 1);
		unsigned char* reused = (unsigned char*)calloc(n0 ? n0 : 1, 1);
		collide_pairs(w, 0, n0, dt, raw, hit, reused);
		append_manifolds(w, 0, n0, raw, hit, reused, mans, &nm);
		free(raw); free(hit); free(reused);
	}
	for (uint32_t k = 0; k < nm; ++k) {
		sgo_body* A = &w->bodies[w->cons[k].a]; sgo_body* B = &w->bodies[w->cons[k].b];
		if (A->is_sensor || B->is_sensor) continue;
		const int actA = body_is_active_for_pairs(A), actB = body_is_active_for_pairs(B);
		if (actA && !actB && B->motion == SGP_MOTION_DYNAMIC) B->can_sleep = -1;   /* mark for wake-up */
		if (actB && !actA && A->motion == SGP_MOTION_DYNAMIC) A->can_sleep = -1;
	}
	/* In-step activation (PhysicsSystem::JobFindCollisions keeps taking bodies from the active list while ProcessBodyPair appends the ones it wakes,
	   so a woken body collides in the step that woke it and wakes what it touches in turn).  Here in one extra round: the bodies marked so far
	   (by a contact above or by a wheel, vehicles_pre_step) take along everything that fell asleep in the same island (sleep_label: sleeping
	   bodies have not moved, so the contacts that made the island are the contacts the cascade would follow), and every woken body is paired
	   with all that was not awake when the step began -- the pairs with awake bodies exist already.  What THOSE contacts wake in turn (two
	   islands that went to sleep apart and touch) is woken too, but meets its other contacts in the next step. */
	if (g_in_step_activation) {
		unsigned char* lab = NULL; uint32_t n_woken = 0;
		for (uint32_t i = 0; i < w->high; ++i) {
			const sgo_body* b = &w->bodies[i];
			if (!b->alive || b->can_sleep != -1) continue;
			if (!lab) lab = (unsigned char*)calloc(w->cap ? w->cap : 1, 1);
			if (SGO_LABEL_CURRENT(w, b->sleep_label)) lab[SGO_LABEL_SLOT(b->sleep_label)] = 1;
		}
		if (lab) {
			for (uint32_t i = 0; i < w->high; ++i) {
				sgo_body* b = &w->bodies[i];
				if (!b->alive || b->is_alias || b->motion != SGP_MOTION_DYNAMIC || b->active) continue;
				if (SGO_LABEL_CURRENT(w, b->sleep_label) && lab[SGO_LABEL_SLOT(b->sleep_label)]) b->can_sleep = -1;
				if (b->can_sleep == -1) ++n_woken;
			}
			free(lab);
		}
		if (n_woken) {
			const float large_r = w->desc.large_body_radius;
			const keyidx* ki = w->bp_ki; const uint32_t n_small = w->bp_n_small; const float cell = w->bp_cell;
			#define WOKEN_CANDIDATE(i_, j_) do { \
				const uint32_t jj_ = (j_); const sgo_body* o_ = &w->bodies[jj_]; \
				if (jj_ != (i_) && o_->alive && !o_->is_alias && !body_is_active_for_pairs(o_) && !(o_->can_sleep == -1 && jj_ < (i_)) && pair_passes_geom(w, (i_), jj_)) push_pair(w, (i_), jj_); \
			} while (0)
			for (uint32_t i = 0; i < w->high; ++i) {
				const sgo_body* b = &w->bodies[i];
				if (!b->alive || b->can_sleep != -1) continue;
				if (body_bounding_radius(b) > large_r) { for (uint32_t j = 0; j < w->high; ++j) WOKEN_CANDIDATE(i, j); continue; }
				const v3 c = v3_scale(v3_add(b->aabb_min, b->aabb_max), 0.5f);
				const int64_t cx = (int64_t)floorf(c.x / cell), cy = (int64_t)floorf(c.y / cell), cz = (int64_t)floorf(c.z / cell);
				for (int64_t dz = -1; dz <= 1; ++dz) for (int64_t dy = -1; dy <= 1; ++dy) for (int64_t dx = -1; dx <= 1; ++dx) {
					const uint64_t key = cell_key(cx + dx, cy + dy, cz + dz);
					for (uint32_t q = lower_bound_key(ki, n_small, key); q < n_small && ki[q].key == key; ++q) WOKEN_CANDIDATE(i, ki[q].idx);
				}
				for (uint32_t l = 0; l < w->n_large; ++l) WOKEN_CANDIDATE(i, w->large[l]);
			}
			#undef WOKEN_CANDIDATE
			const uint32_t n1 = w->n_pairs - n0;
			if (n1) {
				cons_reserve(w, nm + 3 * n1 + 1);
				mans = (sgo_manifold*)realloc(mans, sizeof(sgo_manifold) * (nm + 3 * n1 + 1));
				sgo_manifold* raw = (sgo_manifold*)malloc(sizeof(sgo_manifold) * 3 * n1);
				unsigned char* hit = (unsigned char*)malloc(n1);
				unsigned char* reused = (unsigned char*)calloc(n1, 1);
				collide_pairs(w, n0, n1, dt, raw, hit, reused);
				const uint32_t nm0 = nm;
				append_manifolds(w, n0, n1, raw, hit, reused, mans, &nm);
				free(raw); free(hit); free(reused);
				/* neither body of such a pair was awake when the step began: a contact wakes whichever of the two is dynamic */
				for (uint32_t k = nm0; k < nm; ++k) {
					sgo_body* A = &w->bodies[w->cons[k].a]; sgo_body* B = &w->bodies[w->cons[k].b];
					if (A->is_sensor || B->is_sensor) continue;
					if (A->motion == SGP_MOTION_DYNAMIC && !A->active) A->can_sleep = -1;
					if (B->motion == SGP_MOTION_DYNAMIC && !B->active) B->can_sleep = -1;
				}
			}
		}
	}
	w->n_cons = nm;
	w->stats.num_wake_pairs = w->n_pairs - n0;
	for (uint32_t i = 0; i < w->high; ++i) if (w->bodies[i].alive && w->bodies[i].can_sleep == -1) { w->bodies[i].can_sleep = 0; body_activate(w, i); }

	/* pass 2: constraint properties (independent per constraint: parallel unless contact events must be emitted in order) */
	uint32_t npts = 0, out = 0;
	if (g_threads > 1 && !w->contact_events) {
		unsigned char* keep = (unsigned char*)malloc(nm ? nm : 1);
		#pragma omp parallel for schedule(static, 128) reduction(+:npts)
		for (uint32_t k = 0; k < nm; ++k) { uint32_t np1 = 0; keep[k] = (unsigned char)setup_constraint(w, k, &mans[k], dt, &np1); npts += np1; }
		for (uint32_t k = 0; k < nm; ++k) if (keep[k]) w->cons[out++] = w->cons[k];
		free(keep);
		w->n_cons = out;
		w->stats.num_contact_points = npts;
		free(mans);
		return;
	}
	for (uint32_t k = 0; k < nm; ++k) {
		if (w->contact_events) emit_contact_event(w, &w->cons[k], &mans[k], find_prev(w, w->cons[k].key) != NULL);
		uint32_t np1 = 0;
		if (setup_constraint(w, k, &mans[k], dt, &np1)) { npts += np1; w->cons[out++] = w->cons[k]; }
	}
	w->n_cons = out;
	w->stats.num_contact_points = npts;
	free(mans);
}

/* ------------------------------------------------------------------------------------------------ */
/* deterministic round-based greedy colouring (see header): shared spec with the device code           */

static void colour_constraints(sgo_world* w)
{
	for (uint32_t i = 0; i < w->high; ++i) {
		sgo_body* b = &w->bodies[i];
		b->colour_mask = 0; b->claim[0] = b->claim[1] = ~0ull;
		b->movable_prev = b->movable_cur; b->movable_cur = b->alive && body_movable(b);
		b->chassis = 0;
	}
	/* Colour 0 of a vehicle's chassis is the vehicle's own: its rows are solved before the contacts in every pass (non-contact constraints first,
	   as in PhysicsSystem's solve), and the device solves them in the same launch as the first contact colour -- so no contact of a chassis takes
	   colour 0.  The order a sequential solve visits the rows of any one body in is unchanged by this: vehicle, then contacts by colour. */
	for (uint32_t k = 0; k < w->n_vehicles; ++k) if (w->vehicles[k].alive && w->vehicles[k].body < w->high) w->bodies[w->vehicles[k].body].chassis = 1;
	/* ... and so is colour 0 of a dynamic body under a wheel of an active vehicle: the wheel rows act on it (round 4) */
	for (uint32_t k = 0; k < w->n_vehicles; ++k) {
		const sgo_vehicle* v = &w->vehicles[k];
		if (!v->alive || !v->active) continue;
		for (int i = 0; i < v->num_wheels; ++i) if (v->wheels[i].has_contact && v->wheels[i].ground_dynamic && v->wheels[i].contact_body < w->high) w->bodies[v->wheels[i].contact_body].chassis = 1;
	}
	/* Colour inheritance through the contact cache: a persisted manifold keeps last step's colour when both of its movable
	   bodies were already movable when that colour was chosen (then last step's proper colouring guarantees that no two
	   inheritors sharing a movable body carry the same colour).  Only the other manifolds go through the rounds below. */
	uint32_t remaining = 0, rounds = 0;
	for (uint32_t k = 0; k < w->n_cons; ++k) {
		sgo_constraint* c = &w->cons[k];
		c->colour = -1;
		const sgo_constraint* pc = find_prev(w, c->key);
		sgo_body* A = &w->bodies[c->a]; sgo_body* B = &w->bodies[c->b];
		if (pc && pc->colour >= 0 && pc->colour < SGO_OVERFLOW_COLOUR && (!A->movable_cur || A->movable_prev) && (!B->movable_cur || B->movable_prev)
		    && !(pc->colour == 0 && ((A->movable_cur && A->chassis) || (B->movable_cur && B->chassis)))) {
			c->colour = pc->colour;
			if (A->movable_cur) A->colour_mask |= 1ull << c->colour;
			if (B->movable_cur) B->colour_mask |= 1ull << c->colour;
		} else ++remaining;
	}
	while (remaining) {
		const int cur = rounds & 1, nxt = cur ^ 1;
		/* phase A: claim (atomic min per body; the outcome is order independent) */
		#pragma omp parallel for schedule(static, 512) if (g_threads > 1)
		for (uint32_t k = 0; k < w->n_cons; ++k) {
			sgo_constraint* c = &w->cons[k];
			if (c->colour >= 0) continue;
			sgo_body* A = &w->bodies[c->a]; sgo_body* B = &w->bodies[c->b];
			if (body_movable(A)) { uint64_t cur_v = __atomic_load_n(&A->claim[cur], __ATOMIC_RELAXED); while (c->prio < cur_v && !__atomic_compare_exchange_n(&A->claim[cur], &cur_v, c->prio, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {} }
			if (body_movable(B)) { uint64_t cur_v = __atomic_load_n(&B->claim[cur], __ATOMIC_RELAXED); while (c->prio < cur_v && !__atomic_compare_exchange_n(&B->claim[cur], &cur_v, c->prio, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {} }
		}
		/* phase B: winners take the lowest colour free on both bodies (a body has at most one winner per round) */
		uint32_t won = 0;
		#pragma omp parallel for schedule(static, 512) reduction(+:won) if (g_threads > 1)
		for (uint32_t k = 0; k < w->n_cons; ++k) {
			sgo_constraint* c = &w->cons[k];
			if (c->colour >= 0) continue;
			sgo_body* A = &w->bodies[c->a]; sgo_body* B = &w->bodies[c->b];
			const int ma = body_movable(A), mb = body_movable(B);
			const int win = (!ma || A->claim[cur] == c->prio) && (!mb || B->claim[cur] == c->prio);
			if (win) {
				const uint64_t used = (ma ? A->colour_mask | (uint64_t)A->chassis : 0) | (mb ? B->colour_mask | (uint64_t)B->chassis : 0);
				int col = 0;
				while (col < SGO_OVERFLOW_COLOUR && ((used >> col) & 1)) ++col;
				c->colour = col;
				if (col < SGO_OVERFLOW_COLOUR) {
					if (ma) A->colour_mask |= 1ull << col;
					if (mb) B->colour_mask |= 1ull << col;
				}
				++won;
			}
		}
		remaining -= won;
		/* reset the other claim buffer for the bodies the next round can touch */
		#pragma omp parallel for schedule(static, 512) if (g_threads > 1)
		for (uint32_t k = 0; k < w->n_cons; ++k) {
			sgo_constraint* c = &w->cons[k];
			w->bodies[c->a].claim[nxt] = ~0ull; w->bodies[c->b].claim[nxt] = ~0ull;
		}
		++rounds;
	}
	w->stats.num_colour_rounds = rounds;
}

static const sgo_world* g_sort_world;
static int cmp_order(const void* a, const void* b)
{
	const sgo_constraint* x = &g_sort_world->cons[*(const uint32_t*)a];
	const sgo_constraint* y = &g_sort_world->cons[*(const uint32_t*)b];
	if (x->colour != y->colour) return x->colour - y->colour;
	if (x->prio < y->prio) return -1; if (x->prio > y->prio) return 1;
	return 0;
}

/* ------------------------------------------------------------------------------------------------ */
/* solver (Jolt ContactConstraintManager::WarmStart / SolveVelocityConstraints / SolvePositionConstraints) */

static void apply_impulse(sgo_body* A, sgo_body* B, float im1, sym33 I1, float im2, sym33 I2, v3 r1, v3 r2, v3 axis, float lambda)
{
	if (im1 > 0.0f) {
		A->linv = v3_sub(A->linv, v3_scale(axis, lambda * im1));
		A->angv = v3_sub(A->angv, v3_scale(sym33_mul(I1, v3_cross(r1, axis)), lambda));
	}
	if (im2 > 0.0f) {
		B->linv = v3_add(B->linv, v3_scale(axis, lambda * im2));
		B->angv = v3_add(B->angv, v3_scale(sym33_mul(I2, v3_cross(r2, axis)), lambda));
	}
}

static float axis_jv(const sgo_body* A, const sgo_body* B, v3 r1, v3 r2, v3 axis)
{
	/* each body's share first, then the difference (the device computes the two shares on two lanes) */
	return (v3_dot(axis, A->linv) + v3_dot(v3_cross(r1, axis), A->angv)) - (v3_dot(axis, B->linv) + v3_dot(v3_cross(r2, axis), B->angv));
}

/* Warm start (ContactConstraintManager::WarmStartVelocityConstraints): the cached impulses of a manifold applied to its two bodies.
   Round 6 contract: the impulses of a manifold are SUMMED first -- per point n lam_n (+ t1 lam_t1 + t2 lam_t2 with friction), then over the points the
   linear impulse P and the angular impulses about either centre of mass A1 = sum r1 x j, A2 = sum r2 x j -- and each body receives ONE velocity change,
   v -+ P / m, w -+ I^-1 A.  Jolt applies the parts one after the other (AxisConstraintPart::WarmStart per axis and point); the sum is the same impulse up to
   rounding, and what a body receives from a manifold becomes one 32-byte record (the device's k_setup writes it, its warm start adds a body's records in
   colour order: docs/CONTRACT.md, warm start).  UNVERIFIED: upstream applies the parts sequentially. */
static void warm_start_constraint(sgo_world* w, sgo_constraint* c)
{
	sgo_body* A = &w->bodies[c->a]; sgo_body* B = &w->bodies[c->b];
	const float im1 = body_movable(A) ? A->inv_mass : 0.0f, im2 = body_movable(B) ? B->inv_mass : 0.0f;
	sym33 I1, I2; memset(&I1, 0, sizeof(I1)); memset(&I2, 0, sizeof(I2));
	if (im1 > 0.0f) I1 = world_inv_inertia(quat_to_m33(A->rot), A->inv_inertia);
	if (im2 > 0.0f) I2 = world_inv_inertia(quat_to_m33(B->rot), B->inv_inertia);
	v3 P = V3(0.0f, 0.0f, 0.0f), A1 = V3(0.0f, 0.0f, 0.0f), A2 = V3(0.0f, 0.0f, 0.0f);
	for (int i = 0; i < c->np; ++i) {
		const sgo_point* p = &c->pt[i];
		v3 j = v3_scale(c->n, p->lam_n);
		if (c->friction > 0.0f) { j = v3_add(j, v3_scale(c->t1, p->lam_t1)); j = v3_add(j, v3_scale(c->t2, p->lam_t2)); }
		P = v3_add(P, j);
		A1 = v3_add(A1, v3_cross(p->r1, j));
		A2 = v3_add(A2, v3_cross(p->r2, j));
	}
	if (im1 > 0.0f) { A->linv = v3_sub(A->linv, v3_scale(P, im1)); A->angv = v3_sub(A->angv, sym33_mul(I1, A1)); }
	if (im2 > 0.0f) { B->linv = v3_add(B->linv, v3_scale(P, im2)); B->angv = v3_add(B->angv, sym33_mul(I2, A2)); }
}

static void solve_velocity_constraint(sgo_world* w, sgo_constraint* c)
{
	sgo_body* A = &w->bodies[c->a]; sgo_body* B = &w->bodies[c->b];
	const float im1 = body_movable(A) ? A->inv_mass : 0.0f, im2 = body_movable(B) ? B->inv_mass : 0.0f;
	sym33 I1, I2; memset(&I1, 0, sizeof(I1)); memset(&I2, 0, sizeof(I2));
	if (im1 > 0.0f) I1 = world_inv_inertia(quat_to_m33(A->rot), A->inv_inertia);
	if (im2 > 0.0f) I2 = world_inv_inertia(quat_to_m33(B->rot), B->inv_inertia);
	/* friction first (uses the normal impulse of the previous iteration), then non-penetration */
	if (c->friction > 0.0f) {
		for (int i = 0; i < c->np; ++i) {
			sgo_point* p = &c->pt[i];
			if (p->eff_t1 <= 0.0f && p->eff_t2 <= 0.0f) continue;
			float l1 = p->lam_t1 + p->eff_t1 * axis_jv(A, B, p->r1, p->r2, c->t1);
			float l2 = p->lam_t2 + p->eff_t2 * axis_jv(A, B, p->r1, p->r2, c->t2);
			const float max_f = c->friction * p->lam_n;
			const float tot_sq = l1 * l1 + l2 * l2;
			if (tot_sq > max_f * max_f) { const float sc = max_f / sqrtf(tot_sq); l1 = l1 * sc; l2 = l2 * sc; }
			apply_impulse(A, B, im1, I1, im2, I2, p->r1, p->r2, c->t1, l1 - p->lam_t1); p->lam_t1 = l1;
			apply_impulse(A, B, im1, I1, im2, I2, p->r1, p->r2, c->t2, l2 - p->lam_t2); p->lam_t2 = l2;
		}
	}
	for (int i = 0; i < c->np; ++i) {
		sgo_point* p = &c->pt[i];
		if (p->eff_n <= 0.0f) continue;
		const float jv = axis_jv(A, B, p->r1, p->r2, c->n);
		const float lambda = p->eff_n * (jv - p->bias);
		const float nl = max0f(p->lam_n + lambda);
		apply_impulse(A, B, im1, I1, im2, I2, p->r1, p->r2, c->n, nl - p->lam_n);
		p->lam_n = nl;
	}
}

static void solve_position_constraint(sgo_world* w, sgo_constraint* c)
{
	sgo_body* A = &w->bodies[c->a]; sgo_body* B = &w->bodies[c->b];
	const float im1 = body_movable(A) ? A->inv_mass : 0.0f, im2 = body_movable(B) ? B->inv_mass : 0.0f;
	for (int i = 0; i < c->np; ++i) {
		sgo_point* p = &c->pt[i];
		const m33 RA = quat_to_m33(A->rot), RB = quat_to_m33(B->rot);
		const v3 p1 = v3_add(A->pos, m33_mul(RA, p->local1));
		const v3 p2 = v3_add(B->pos, m33_mul(RB, p->local2));
		float sep = v3_dot(v3_sub(p2, p1), c->n) + w->st.penetration_slop;
		if (sep < 0.0f) {
			sep = fmaxf(sep, -w->st.max_penetration_distance);
			const v3 mid = v3_scale(v3_add(p1, p2), 0.5f);
			const v3 r1 = v3_sub(mid, A->pos), r2 = v3_sub(mid, B->pos);
			sym33 I1, I2; memset(&I1, 0, sizeof(I1)); memset(&I2, 0, sizeof(I2));
			if (im1 > 0.0f) I1 = world_inv_inertia(RA, A->inv_inertia);
			if (im2 > 0.0f) I2 = world_inv_inertia(RB, B->inv_inertia);
			const float eff = axis_eff_mass(im1, I1, r1, im2, I2, r2, c->n);
			if (eff <= 0.0f) continue;
			const float lambda = -eff * w->st.baumgarte * sep;
			if (im1 > 0.0f) {
				A->pos = v3_sub(A->pos, v3_scale(c->n, lambda * im1));
				A->rot = quat_add_rotation_step(A->rot, v3_scale(sym33_mul(I1, v3_cross(r1, c->n)), -lambda));
			}
			if (im2 > 0.0f) {
				B->pos = v3_add(B->pos, v3_scale(c->n, lambda * im2));
				B->rot = quat_add_rotation_step(B->rot, v3_scale(sym33_mul(I2, v3_cross(r2, c->n)), lambda));
			}
		}
	}
}

/* ------------------------------------------------------------------------------------------------ */
/* islands (union-find, root = smallest body id) and sleeping (Body::UpdateSleepStateInternal)        */

/* the priority the device's union-find hooks by (a bijection of the id: no ties) */
static uint32_t uf_prio(uint32_t x) { uint32_t h = x * 0x9E3779B1u; h ^= h >> 15; h *= 0x85EBCA6Bu; h ^= h >> 13; return h; }
static uint32_t uf_find(sgo_world* w, uint32_t x)
{
	while ((uint32_t)w->bodies[x].island != x) { w->bodies[x].island = w->bodies[w->bodies[x].island].island; x = (uint32_t)w->bodies[x].island; }
	return x;
}

static void update_sleeping(sgo_world* w, float dt)
{
	const float max_movement = w->st.point_velocity_sleep_threshold * w->st.time_before_sleep;
	for (uint32_t i = 0; i < w->high; ++i) { w->bodies[i].island = (int)i; }
	for (uint32_t k = 0; k < w->n_cons; ++k) {
		const sgo_constraint* c = &w->cons[k];
		if (!body_movable(&w->bodies[c->a]) || !body_movable(&w->bodies[c->b])) continue;
		const uint32_t ra = uf_find(w, c->a), rb = uf_find(w, c->b);
		if (ra < rb) w->bodies[rb].island = (int)ra; else if (rb < ra) w->bodies[ra].island = (int)rb;
	}
	/* a vehicle links its chassis with the dynamic bodies under its wheels (VehicleConstraint::BuildIslands) */
	for (uint32_t k = 0; k < w->n_vehicles; ++k) {
		const sgo_vehicle* v = &w->vehicles[k];
		if (!v->alive || !v->active || !live(w, v->body)) continue;
		for (int i = 0; i < v->num_wheels; ++i) {
			const sgo_wheel* wh = &v->wheels[i];
			if (!wh->has_contact || !wh->ground_dynamic || !live(w, wh->contact_body)) continue;
			if (!body_movable(&w->bodies[v->body]) || !body_movable(&w->bodies[wh->contact_body])) continue;
			const uint32_t ra = uf_find(w, v->body), rb = uf_find(w, wh->contact_body);
			if (ra < rb) w->bodies[rb].island = (int)ra; else if (rb < ra) w->bodies[ra].island = (int)rb;
		}
	}
	/* per-body sleep test */
	for (uint32_t i = 0; i < w->high; ++i) {
		sgo_body* b = &w->bodies[i];
		b->can_sleep = 1;
		if (!b->alive || !body_movable(b)) continue;
		if (!b->allow_sleep || !w->st.allow_sleeping) { b->can_sleep = 0; continue; }
		v3 pts[3];
		body_sleep_points(b, pts);
		int reset = 0;
		for (int k = 0; k < 3; ++k) {
			/* Sphere::EncapsulatePoint */
			const v3 d = v3_sub(pts[k], b->sleep_c[k]);
			const float d2 = v3_len_sq(d);
			if (d2 > b->sleep_r[k] * b->sleep_r[k]) {
				const float dl = sqrtf(d2);
				const float nr = 0.5f * (b->sleep_r[k] + dl);
				b->sleep_c[k] = v3_add(b->sleep_c[k], v3_scale(d, (nr - b->sleep_r[k]) / dl));
				b->sleep_r[k] = nr;
			}
			if (b->sleep_r[k] > max_movement) reset = 1;
		}
		if (reset) {
			for (int k = 0; k < 3; ++k) { b->sleep_c[k] = pts[k]; b->sleep_r[k] = 0.0f; }
			b->sleep_timer = 0.0f; b->can_sleep = 0;
		} else {
			b->sleep_timer += dt;
			b->can_sleep = b->sleep_timer >= w->st.time_before_sleep;
		}
	}
	/* island AND-reduction into the root's colour_mask scratch (1 = every member can sleep) */
	for (uint32_t i = 0; i < w->high; ++i) w->bodies[i].colour_mask = 1;
	for (uint32_t i = 0; i < w->high; ++i) {
		sgo_body* b = &w->bodies[i];
		if (!b->alive || !body_movable(b)) continue;
		if (!b->can_sleep) w->bodies[uf_find(w, i)].colour_mask = 0;
	}
	/* an island that goes to sleep is remembered by its members (sleep_label = the member the device's union-find ends with as the root: the one
	   with the lowest uf_prio): what wakes one of them later wakes all of them in that step (find_contacts) */
	uint32_t* best = NULL;
	for (uint32_t i = 0; i < w->high; ++i) {
		const sgo_body* b = &w->bodies[i];
		if (!b->alive || !body_movable(b)) continue;
		const uint32_t r = uf_find(w, i);
		if (!w->bodies[r].colour_mask) continue;
		if (!best) { best = (uint32_t*)malloc(sizeof(uint32_t) * w->high); for (uint32_t k = 0; k < w->high; ++k) best[k] = SGP_INVALID_ID; }
		if (best[r] == SGP_INVALID_ID || uf_prio(i) < uf_prio(best[r])) best[r] = i;
	}
	for (uint32_t i = 0; i < w->high; ++i) {
		sgo_body* b = &w->bodies[i];
		if (!b->alive || !body_movable(b)) continue;
		const uint32_t r = uf_find(w, i);
		if (w->bodies[r].colour_mask) {
			b->active = 0; b->linv = V3(0, 0, 0); b->angv = V3(0, 0, 0);
			b->sleep_label = SGO_LABEL(best[r], w->bodies[best[r]].slot_gen & 0x7Fu);
			push_body_event(w, SGP_EVENT_DEACTIVATED, i);
		}
	}
	free(best);
	/* kinematic bodies stay active while they move */
	for (uint32_t i = 0; i < w->high; ++i) {
		sgo_body* b = &w->bodies[i];
		if (b->alive && b->motion == SGP_MOTION_KINEMATIC && b->active && v3_len_sq(b->linv) == 0.0f && v3_len_sq(b->angv) == 0.0f) {
			b->active = 0; push_body_event(w, SGP_EVENT_DEACTIVATED, i);
		}
	}
}

/* ------------------------------------------------------------------------------------------------ */
/* buoyancy sweep, PhysicsWorld.cpp:1367-1442 (Body::GetSubmergedVolume + Body::ApplyBuoyancyImpulse)  */

/* Volume and centroid of the part of a convex polyhedron (box) below the plane z = wz, by slicing the 8
   corners: uses the exact tetrahedral decomposition of the clipped box. */
static void box_submerged_h(v3 h, m33 R, float posz, float wz, float* vol_out, v3* centroid_out)
{
	/* Work in the box frame: plane n.x = d with n = R^T (0,0,1), d = wz - pos.z ; submerged part: n.x <= d. */
	const v3 n = m33_tmul(R, V3(0.0f, 0.0f, 1.0f));
	const float d = wz - posz;
	/* Clip the 6 faces (quads) by the half space and add the cap polygon; accumulate signed tetrahedra from the origin. */
	float vol = 0.0f; v3 cen = V3(0, 0, 0);
	v3 cap[24]; int ncap = 0;
	for (int ax = 0; ax < 3; ++ax) for (int sg = -1; sg <= 1; sg += 2) {
		const int u = (ax + 1) % 3, v = (ax + 2) % 3;
		v3 q[4];
		const float su[4] = { 1, -1, -1, 1 }, sv[4] = { 1, 1, -1, -1 };
		for (int k = 0; k < 4; ++k) {
			v3 p = V3(0, 0, 0);
			v3_set(&p, ax, (float)sg * v3_get(h, ax));
			/* orient counter-clockwise seen from outside */
			const int kk = sg > 0 ? k : 3 - k;
			v3_set(&p, u, su[kk] * v3_get(h, u)); v3_set(&p, v, sv[kk] * v3_get(h, v));
			q[k] = p;
		}
		v3 poly[8]; int np = 0;
		for (int k = 0; k < 4; ++k) {
			const v3 a = q[k], c = q[(k + 1) % 4];
			const float da = v3_dot(n, a) - d, dc = v3_dot(n, c) - d;
			if (da <= 0.0f) poly[np++] = a;
			if ((da <= 0.0f) != (dc <= 0.0f)) {
				const float t = da / (da - dc);
				const v3 x = v3_add(a, v3_scale(v3_sub(c, a), t));
				poly[np++] = x;
				if (ncap < 24) cap[ncap++] = x;
			}
		}
		for (int k = 1; k + 1 < np; ++k) {
			const float tv = v3_dot(poly[0], v3_cross(poly[k], poly[k + 1])) / 6.0f;
			vol += tv;
			cen = v3_add(cen, v3_scale(v3_add(v3_add(poly[0], poly[k]), poly[k + 1]), tv * 0.25f));
		}
	}
	/* cap polygon lies in the plane n.x = d: its tetrahedra with the origin have volume (d/3)*area each; order the
	   cap points by angle around their mean */
	if (ncap >= 3) {
		v3 mean = V3(0, 0, 0);
		for (int k = 0; k < ncap; ++k) mean = v3_add(mean, cap[k]);
		mean = v3_scale(mean, 1.0f / (float)ncap);
		const v3 e1 = v3_normalized_perpendicular(n), e2 = v3_cross(n, e1);
		float ang[24];
		for (int k = 0; k < ncap; ++k) {
			/* monotone pseudo-angle in (-2, 2]: no libm, so the device code orders the points identically */
			const v3 r = v3_sub(cap[k], mean);
			const float dx = v3_dot(r, e1), dy = v3_dot(r, e2);
			const float den = fabsf(dx) + fabsf(dy);
			const float pa = den > 0.0f ? 1.0f - dx / den : 0.0f;
			ang[k] = dy < 0.0f ? -pa : pa;
		}
		for (int i = 1; i < ncap; ++i) { const float a = ang[i]; const v3 p = cap[i]; int j = i - 1; while (j >= 0 && ang[j] > a) { ang[j + 1] = ang[j]; cap[j + 1] = cap[j]; --j; } ang[j + 1] = a; cap[j + 1] = p; }
		for (int k = 0; k < ncap; ++k) {
			const v3 a = cap[k], c = cap[(k + 1) % ncap];
			const float tv = v3_dot(mean, v3_cross(a, c)) / 6.0f;
			vol += tv;
			cen = v3_add(cen, v3_scale(v3_add(v3_add(mean, a), c), tv * 0.25f));
		}
	}
	*vol_out = vol;
	*centroid_out = vol > 1.0e-12f ? m33_mul(R, v3_scale(cen, 1.0f / vol)) : V3(0, 0, 0);   /* relative to COM, world axes */
}

static void sphere_cap_submerged(float r, float depth_of_centre /* wz - centre.z */, float* vol_out, float* cz_out)
{
	/* submerged height h in [0, 2r] measured from the bottom of the sphere */
	const float h = clampf(depth_of_centre + r, 0.0f, 2.0f * r);
	const float pi = 3.14159265358979323846f;
	const float vol = pi * h * h * (3.0f * r - h) / 3.0f;
	/* centroid of a spherical cap of height h, measured from the sphere centre, pointing down */
	float cz = 0.0f;
	if (h > 0.0f) { const float k = 2.0f * r - h; cz = -(3.0f * k * k) / (4.0f * (3.0f * r - h)); }
	*vol_out = vol; *cz_out = cz;
}

/* ConvexHullShape::GetSubmergedVolume: the exact volume and centre of the part of the polyhedron under the plane.  Every face polygon is clipped
   to the half space and fanned into tetrahedra whose apex lies IN the plane (the point of the plane closest to the centre of mass), so that the
   cut surface itself contributes nothing.  Hull frame = body frame, origin = centre of mass. */
static void hull_submerged(const sgo_hull* hl, m33 R, float posz, float wz, float* vol_out, v3* centroid_out)
{
	const v3 n = m33_tmul(R, V3(0.0f, 0.0f, 1.0f));
	const float d = wz - posz;
	float lo = 3.4e38f, hi = -3.4e38f;
	for (int i = 0; i < hl->nv; ++i) { const float t = v3_dot(n, hl->verts[i]); lo = fminf(lo, t); hi = fmaxf(hi, t); }
	if (lo >= d) { *vol_out = 0.0f; *centroid_out = V3(0, 0, 0); return; }
	if (hi <= d) { *vol_out = hl->volume; *centroid_out = V3(0, 0, 0); return; }
	const v3 apex = v3_scale(n, d);
	float vol = 0.0f; v3 cen = V3(0, 0, 0);
	for (int f = 0; f < hl->nf; ++f) {
		const int b0 = hl->face_start[f], cnt = hl->face_start[f + 1] - b0;
		/* (the clipped polygon is fanned as its corners come: first corner, previous corner, this corner -- a face holds up to 256 of them) */
		v3 p0 = V3(0, 0, 0), pp = V3(0, 0, 0); int np = 0;
		for (int k = 0; k < cnt; ++k) {
			const v3 a = hl->verts[hl->face_idx[b0 + k]], c = hl->verts[hl->face_idx[b0 + (k + 1) % cnt]];
			const float da = v3_dot(n, a) - d, dc = v3_dot(n, c) - d;
			v3 q[2]; int nq = 0;
			if (da <= 0.0f) q[nq++] = v3_sub(a, apex);
			if ((da <= 0.0f) != (dc <= 0.0f)) { const float t = da / (da - dc); q[nq++] = v3_sub(v3_add(a, v3_scale(v3_sub(c, a), t)), apex); }
			for (int i = 0; i < nq; ++i) {
				if (np == 0) p0 = q[i];
				else if (np >= 2) {
					const float tv = v3_dot(p0, v3_cross(pp, q[i])) / 6.0f;
					vol += tv;
					cen = v3_add(cen, v3_scale(v3_add(v3_add(p0, pp), q[i]), tv * 0.25f));
				}
				pp = q[i]; ++np;
			}
		}
	}
	*vol_out = vol;
	*centroid_out = vol > 1.0e-12f ? m33_mul(R, v3_add(apex, v3_scale(cen, 1.0f / vol))) : V3(0, 0, 0);
}

/* Shape::GetSubmergedVolume as Jolt's shapes implement it: box and convex hull exactly (polyhedra), sphere by the cap formula, and every
   other convex shape -- here the capsule -- through ConvexShape::GetSubmergedVolume, which stands the shape's LOCAL BOUNDING BOX in for it:
   total volume = that box's, submerged part = the box's part under the plane.  (The caller's buoyancy factor still comes from
   Shape::GetVolume, PhysicsWorld.cpp:1387, so what matters is the box's submerged FRACTION; the reported volume is the box's, as the
   reference would get it back.)  UNVERIFIED: upstream. */
static void submerged_volume(const sgo_body* b, float wz, float* total, float* sub, v3* rel_cob)
{
	*total = shape_volume_h(b->shape_type, b->shape, b->hull);
	if (b->shape_type == SGP_SHAPE_BOX) { box_submerged_h(V3(b->shape[0], b->shape[1], b->shape[2]), quat_to_m33(b->rot), b->pos.z, wz, sub, rel_cob); return; }
	if (b->shape_type == SGP_SHAPE_HULL && b->hull) { hull_submerged(b->hull, quat_to_m33(b->rot), b->pos.z, wz, sub, rel_cob); return; }
	if (b->shape_type == SGP_SHAPE_CAPSULE) {
		const v3 h = V3(b->shape[0], b->shape[0], b->shape[1] + b->shape[0]);      /* capsule along z: radius, half height of the cylinder */
		*total = 8.0f * h.x * h.y * h.z;
		box_submerged_h(h, quat_to_m33(b->rot), b->pos.z, wz, sub, rel_cob);
		return;
	}
	if (b->shape_type == SGP_SHAPE_SPHERE) {
		float cz; sphere_cap_submerged(b->shape[0], wz - b->pos.z, sub, &cz);
		*rel_cob = V3(0.0f, 0.0f, cz);
		return;
	}
	/* anything else (none today): the fraction of its AABB height under water */
	{
		const float zmin = b->aabb_min.z, zmax = b->aabb_max.z;
		const float f = clampf((wz - zmin) / (zmax - zmin), 0.0f, 1.0f);
		*sub = *total * f;
		*rel_cob = V3(0.0f, 0.0f, (zmin + 0.5f * f * (zmax - zmin)) - b->pos.z);
	}
}

static void buoyancy_sweep(sgo_world* w, float dt)
{
	for (uint32_t i = 0; i < w->high; ++i) {
		sgo_body* b = &w->bodies[i];
		if (!b->alive || !b->active || b->motion != SGP_MOTION_DYNAMIC) continue;          /* :1377 */
		if (b->aabb_min.z < w->water_z) {                                                   /* :1379 */
			const float fluid_density = 1020.0f;                                            /* :1381 */
			float total, sub; v3 rc;
			submerged_volume(b, w->water_z, &total, &sub, &rc);
			const float buoyancy = fluid_density * shape_volume_h(b->shape_type, b->shape, b->hull) / b->mass;      /* :1387: Shape::GetVolume, the real one */
			int applied = 0;
			if (sub > 0.0f) {
				/* Body::ApplyBuoyancyImpulse */
				const float inv_mass = b->inv_mass;
				const float rho = buoyancy / (total * inv_mass);
				const v3 g = V3(0.0f, 0.0f, -9.81f);                                         /* :1407 */
				const v3 buoy_imp = v3_scale(g, -rho * sub * b->gravity_factor * dt);
				const v3 cob_vel = v3_add(b->linv, v3_cross(b->angv, rc));
				const v3 rel = v3_neg(cob_vel);                                              /* fluid velocity 0, :1406 */
				const float lin_drag = b->zero_lin_drag ? 0.0f : 0.1f;                      /* :1404 */
				const v3 size = v3_scale(shape_local_half_h(b->shape_type, b->shape, b->hull), 2.0f);
				const m33 R = quat_to_m33(b->rot);
				const v3 lrel = m33_tmul(R, rel);
				const float rl2 = v3_len_sq(lrel);
				v3 drag_imp = V3(0, 0, 0);
				if (rl2 > 1.0e-12f) {
					const float rl = sqrtf(rl2);
					const v3 dirl = v3_scale(v3_abs(lrel), 1.0f / rl);
					const float area = (sub / total) * (dirl.x * size.y * size.z + dirl.y * size.x * size.z + dirl.z * size.x * size.y);
					float dv = 0.5f * rho * rl2 * lin_drag * area * dt * inv_mass;
					if (dv > rl) dv = rl;
					drag_imp = v3_scale(rel, dv / (rl * inv_mass));
				}
				const v3 dlin = v3_scale(v3_add(drag_imp, buoy_imp), inv_mass);
				const float l = (size.x + size.y + size.z) / 3.0f;
				const float ang_drag = 3.0f;                                                 /* :1405 */
				const v3 drag_ang_imp = v3_scale(b->angv, -ang_drag * sub / total * dt * (l * l) / inv_mass);
				const sym33 Iw = world_inv_inertia(R, b->inv_inertia);
				v3 ddrag = sym33_mul(Iw, drag_ang_imp);
				if (v3_len_sq(ddrag) > v3_len_sq(b->angv)) ddrag = v3_neg(b->angv);
				const v3 dang = v3_add(ddrag, sym33_mul(Iw, v3_cross(rc, v3_add(buoy_imp, drag_imp))));
				b->linv = v3_add(b->linv, dlin);
				b->angv = v3_add(b->angv, dang);
				applied = 1;
			}
			if (applied) {
				if (!b->underwater) { push_body_event(w, SGP_EVENT_ENTERED_WATER, i); b->underwater = 1; }  /* :1414-1420 */
				b->submerged = sub;                                                                       /* :1422 */
			} else { b->underwater = 0; b->submerged = 0.0f; }                                              /* :1426-1427 */
		} else if (b->underwater) { b->underwater = 0; b->submerged = 0.0f; }                               /* :1432-1436 */
	}
}

/* ------------------------------------------------------------------------------------------------ */
/* think(dt)                                                                                        */


/* ------------------------------------------------------------------------------------------------ */
/* wheeled vehicles: world-side glue around sgo_vehicle.h                                              */

SGO_API void sgo_default_vehicle_desc(sgp_vehicle_desc* d)
{
	memset(d, 0, sizeof(*d));
	d->body = SGP_INVALID_ID;
	d->num_wheels = 4;
	for (int i = 0; i < 4; ++i) {
		sgp_wheel_desc* w = &d->wheels[i];
		const int front = i < 2, left = (i % 2) == 0;
		w->position[0] = left ? -0.8f : 0.8f; w->position[1] = front ? 1.3f : -1.3f; w->position[2] = 0.15f;
		w->suspension_dir[2] = -1.0f; w->steering_axis[2] = 1.0f; w->wheel_up[2] = 1.0f; w->wheel_forward[1] = 1.0f;
		w->suspension_min_length = 0.2f; w->suspension_max_length = 0.5f; w->suspension_preload = 0.0f;   /* Scripting.cpp:326-330 */
		w->spring_frequency = 2.0f; w->spring_damping = 0.5f;                                            /* :335-339 */
		w->radius = 0.42f; w->width = 0.16f;                                                             /* :320-324 */
		w->inertia = 0.9f; w->angular_damping = 0.2f;
		w->max_steer_angle = front ? 0.78525f : 0.0f;                                                    /* :342, CarPhysics.cpp:127,153 */
		w->max_brake_torque = 1500.0f; w->max_handbrake_torque = front ? 0.0f : 4000.0f;                 /* :347-348, CarPhysics.cpp:129,155 */
		const float lf[3][2] = { { 0.0f, 0.0f }, { 0.06f, 1.2f }, { 0.2f, 1.0f } };
		const float tf[3][2] = { { 0.0f, 0.0f }, { 3.0f, 1.2f }, { 20.0f, 1.0f } };
		memcpy(w->longitudinal_friction, lf, sizeof(lf)); memcpy(w->lateral_friction, tf, sizeof(tf));
	}
	d->up[2] = 1.0f; d->forward[1] = 1.0f;
	d->cast_radius = 0.08f;                                                                            /* CarPhysics.cpp:62 */
	d->max_slope_angle = 80.0f * 3.14159265358979323846f / 180.0f;
	d->engine_max_torque = 500.0f; d->engine_min_rpm = 1000.0f; d->engine_max_rpm = 6000.0f; d->engine_inertia = 0.5f; d->engine_angular_damping = 0.2f;
	const float ec[3][2] = { { 0.0f, 0.8f }, { 0.66f, 1.0f }, { 1.0f, 0.8f } };
	memcpy(d->engine_torque_curve, ec, sizeof(ec));
	d->num_gears = 5; d->num_reverse_gears = 1;
	const float gr[5] = { 2.66f, 1.78f, 1.3f, 1.0f, 0.74f };
	memcpy(d->gear_ratios, gr, sizeof(gr)); d->reverse_gear_ratios[0] = -2.9f;
	d->switch_time = 0.5f; d->clutch_release_time = 0.3f; d->switch_latency = 0.5f; d->shift_up_rpm = 4000.0f; d->shift_down_rpm = 2000.0f; d->clutch_strength = 10.0f;
	d->num_differentials = 1;                                                                          /* front wheel drive, CarPhysics.cpp:191-194 */
	d->differentials[0].left_wheel = 0; d->differentials[0].right_wheel = 1;
	d->differentials[0].differential_ratio = 3.42f; d->differentials[0].left_right_split = 0.5f; d->differentials[0].limited_slip_ratio = 1.4f; d->differentials[0].engine_torque_ratio = 1.0f;
	d->differentials[1] = d->differentials[0]; d->differentials[1].left_wheel = 2; d->differentials[1].right_wheel = 3;
	d->differential_limited_slip_ratio = 1.4f;
	d->num_anti_roll_bars = 2;                                                                         /* CarPhysics.cpp:217-221 */
	d->anti_roll_bars[0].left_wheel = 0; d->anti_roll_bars[0].right_wheel = 1; d->anti_roll_bars[0].stiffness = 1000.0f;
	d->anti_roll_bars[1].left_wheel = 2; d->anti_roll_bars[1].right_wheel = 3; d->anti_roll_bars[1].stiffness = 1000.0f;
	d->controller_type = SGP_VEHICLE_CONTROLLER_WHEELED;
	d->max_lean_angle = 45.0f * 3.14159265358979323846f / 180.0f; d->lean_spring_constant = 5000.0f; d->lean_spring_damping = 1000.0f;   /* JPH::MotorcycleControllerSettings defaults */
	d->lean_spring_integration_coefficient = 0.0f; d->lean_spring_integration_decay = 4.0f; d->lean_smoothing_factor = 0.8f; d->lean_steering_limit = 1;
}

static v3 v3_from(const float* p) { return V3(p[0], p[1], p[2]); }

static int vehicle_desc_valid(const sgp_vehicle_desc* d)
{
	if (d->num_wheels < 1 || d->num_wheels > SGP_MAX_WHEELS) return 0;
	if (d->num_gears < 1 || d->num_gears > SGP_MAX_GEARS || d->num_reverse_gears < 1 || d->num_reverse_gears > SGP_MAX_GEARS) return 0;
	if (d->num_differentials > 2 || d->num_anti_roll_bars > 2) return 0;
	for (uint32_t k = 0; k < d->num_differentials; ++k) {
		if (d->differentials[k].left_wheel >= (int)d->num_wheels || d->differentials[k].right_wheel >= (int)d->num_wheels) return 0;
		if (!(d->differentials[k].limited_slip_ratio > 1.0f)) return 0;
	}
	for (uint32_t k = 0; k < d->num_anti_roll_bars; ++k) {
		const sgp_anti_roll_bar_desc* r = &d->anti_roll_bars[k];
		if (r->left_wheel < 0 || r->right_wheel < 0 || r->left_wheel >= (int)d->num_wheels || r->right_wheel >= (int)d->num_wheels) return 0;
	}
	for (uint32_t i = 0; i < d->num_wheels; ++i) {
		const sgp_wheel_desc* w = &d->wheels[i];
		if (!(w->radius > 0.0f) || !(w->inertia > 0.0f) || !(w->suspension_max_length >= w->suspension_min_length) || !(w->suspension_min_length >= 0.0f)) return 0;
	}
	if (!(d->engine_inertia > 0.0f) || !(d->engine_max_rpm > 0.0f) || !(d->clutch_release_time > 0.0f) || !(d->differential_limited_slip_ratio > 1.0f)) return 0;
	if (d->controller_type != SGP_VEHICLE_CONTROLLER_WHEELED && d->controller_type != SGP_VEHICLE_CONTROLLER_MOTORCYCLE) return 0;
	if (d->controller_type == SGP_VEHICLE_CONTROLLER_MOTORCYCLE && !(d->max_lean_angle > 0.0f && d->max_lean_angle < 1.5f)) return 0;
	return 1;
}

static void vehicle_from_desc(sgo_vehicle* v, const sgp_vehicle_desc* d)
{
	memset(v, 0, sizeof(*v));
	v->body = d->body; v->alive = 1; v->num_wheels = (int)d->num_wheels;
	for (int i = 0; i < v->num_wheels; ++i) {
		sgo_wheel* w = &v->wheels[i]; const sgp_wheel_desc* s = &d->wheels[i];
		w->position = v3_from(s->position); w->suspension_dir = v3_from(s->suspension_dir); w->steering_axis = v3_from(s->steering_axis);
		w->wheel_up = v3_from(s->wheel_up); w->wheel_forward = v3_from(s->wheel_forward);
		w->sus_min = s->suspension_min_length; w->sus_max = s->suspension_max_length; w->sus_preload = s->suspension_preload;
		w->spring_freq = s->spring_frequency; w->spring_damp = s->spring_damping;
		w->radius = s->radius; w->width = s->width; w->inertia = s->inertia; w->ang_damping = s->angular_damping;
		w->max_steer = s->max_steer_angle; w->max_brake_torque = s->max_brake_torque; w->max_handbrake_torque = s->max_handbrake_torque;
		memcpy(w->long_fric, s->longitudinal_friction, sizeof(w->long_fric)); memcpy(w->lat_fric, s->lateral_friction, sizeof(w->lat_fric));
		w->suspension_length = w->sus_max; w->contact_body = SGP_INVALID_ID;
	}
	v->up = v3_from(d->up); v->forward = v3_from(d->forward);
	v->cast_radius = d->cast_radius;
	v->tester = d->collision_tester == SGP_VEHICLE_TESTER_CYLINDER ? SGP_VEHICLE_TESTER_CYLINDER : SGP_VEHICLE_TESTER_SPHERE;
	{ float sn, cs; sgp_sincos_poly(d->max_slope_angle, &sn, &cs); v->cos_max_slope = cs; }
	v->engine_max_torque = d->engine_max_torque; v->engine_min_rpm = d->engine_min_rpm; v->engine_max_rpm = d->engine_max_rpm;
	v->engine_inertia = d->engine_inertia; v->engine_ang_damping = d->engine_angular_damping;
	memcpy(v->engine_curve, d->engine_torque_curve, sizeof(v->engine_curve));
	v->engine_rpm = d->engine_min_rpm;
	v->num_gears = (int)d->num_gears; v->num_reverse_gears = (int)d->num_reverse_gears;
	memcpy(v->gear_ratios, d->gear_ratios, sizeof(v->gear_ratios)); memcpy(v->reverse_gear_ratios, d->reverse_gear_ratios, sizeof(v->reverse_gear_ratios));
	v->switch_time = d->switch_time; v->clutch_release_time = d->clutch_release_time; v->switch_latency = d->switch_latency;
	v->shift_up_rpm = d->shift_up_rpm; v->shift_down_rpm = d->shift_down_rpm; v->clutch_strength = d->clutch_strength;
	v->current_gear = 0; v->clutch_friction = 1.0f;
	v->num_differentials = (int)d->num_differentials;
	for (int k = 0; k < v->num_differentials; ++k) {
		const sgp_differential_desc* s = &d->differentials[k];
		v->differentials[k].left = s->left_wheel; v->differentials[k].right = s->right_wheel; v->differentials[k].ratio = s->differential_ratio;
		v->differentials[k].left_right_split = s->left_right_split; v->differentials[k].limited_slip_ratio = s->limited_slip_ratio;
		v->differentials[k].engine_torque_ratio = s->engine_torque_ratio;
	}
	v->differential_limited_slip_ratio = d->differential_limited_slip_ratio;
	v->num_anti_roll_bars = (int)d->num_anti_roll_bars;
	for (int k = 0; k < v->num_anti_roll_bars; ++k) {
		v->anti_roll_bars[k].left = d->anti_roll_bars[k].left_wheel; v->anti_roll_bars[k].right = d->anti_roll_bars[k].right_wheel;
		v->anti_roll_bars[k].stiffness = d->anti_roll_bars[k].stiffness;
	}
	v->is_motorcycle = d->controller_type == SGP_VEHICLE_CONTROLLER_MOTORCYCLE;
	v->lean_enabled = v->is_motorcycle; v->lean_steering_limit = d->lean_steering_limit != 0;
	v->max_lean_angle = d->max_lean_angle;
	{ float sn, cs; sgp_sincos_poly(d->max_lean_angle, &sn, &cs); v->tan_max_lean = sn / cs; }
	v->lean_spring_constant = d->lean_spring_constant; v->lean_spring_damping = d->lean_spring_damping;
	v->lean_integration_coefficient = d->lean_spring_integration_coefficient; v->lean_integration_decay = d->lean_spring_integration_decay;
	v->lean_smoothing = d->lean_smoothing_factor;
	v->target_lean = V3(0.0f, 0.0f, 1.0f);
}

SGO_API int sgo_vehicle_create(sgo_world* w, const sgp_vehicle_desc* d, uint32_t* id_out)
{
	if (!w || !d || !id_out) return SGP_ERR_INVALID;
	if (!live(w, d->body) || w->bodies[d->body].motion != SGP_MOTION_DYNAMIC) return SGP_ERR_BAD_ID;
	if (!vehicle_desc_valid(d)) return SGP_ERR_INVALID;
	uint32_t id = w->n_vehicles;
	for (uint32_t k = 0; k < w->n_vehicles; ++k) if (!w->vehicles[k].alive) { id = k; break; }     /* lowest free slot */
	if (id == w->n_vehicles) {
		if (w->n_vehicles == w->cap_vehicles) { w->cap_vehicles = w->cap_vehicles ? w->cap_vehicles * 2 : 16; w->vehicles = (sgo_vehicle*)realloc(w->vehicles, sizeof(sgo_vehicle) * w->cap_vehicles); }
		w->n_vehicles++;
	}
	vehicle_from_desc(&w->vehicles[id], d);
	w->vehicles[id].gravity_len = v3_len(w->gravity);
	*id_out = id;
	return SGP_OK;
}

static int vehicle_live(sgo_world* w, uint32_t id) { return w && id < w->n_vehicles && w->vehicles[id].alive; }

SGO_API int sgo_vehicle_destroy(sgo_world* w, uint32_t id) { if (!vehicle_live(w, id)) return SGP_ERR_BAD_ID; w->vehicles[id].alive = 0; return SGP_OK; }

SGO_API int sgo_vehicle_set_inputs(sgo_world* w, uint32_t first, uint32_t n, const sgp_vehicle_input* in)
{
	for (uint32_t k = 0; k < n; ++k) {
		if (!vehicle_live(w, first + k)) return SGP_ERR_BAD_ID;
		sgo_vehicle* v = &w->vehicles[first + k];
		v->in_forward = clampf(in[k].forward, -1.0f, 1.0f); v->in_right = clampf(in[k].right, -1.0f, 1.0f);
		v->in_brake = clampf(in[k].brake, 0.0f, 1.0f); v->in_handbrake = clampf(in[k].hand_brake, 0.0f, 1.0f);
		/* "On user input, assure that the car is active" (CarPhysics.cpp:362-363) */
		if ((v->in_forward != 0.0f || v->in_right != 0.0f || v->in_brake != 0.0f || v->in_handbrake != 0.0f) && live(w, v->body)) body_activate(w, v->body);
	}
	return SGP_OK;
}
SGO_API int sgo_vehicle_set_input(sgo_world* w, uint32_t id, const sgp_vehicle_input* in) { return sgo_vehicle_set_inputs(w, id, 1, in); }

static void vec_out(float* o, v3 v) { o[0] = v.x; o[1] = v.y; o[2] = v.z; }
SGO_API int sgo_vehicle_get_states(sgo_world* w, uint32_t first, uint32_t n, sgp_vehicle_state* out)
{
	for (uint32_t k = 0; k < n; ++k) {
		if (!vehicle_live(w, first + k)) return SGP_ERR_BAD_ID;
		const sgo_vehicle* v = &w->vehicles[first + k];
		sgp_vehicle_state* s = &out[k];
		memset(s, 0, sizeof(*s));
		for (int i = 0; i < v->num_wheels; ++i) {
			const sgo_wheel* wh = &v->wheels[i]; sgp_wheel_state* ws = &s->wheels[i];
			ws->suspension_length = wh->suspension_length; ws->steer_angle = wh->steer_angle; ws->rotation_angle = wh->angle; ws->angular_velocity = wh->angular_velocity;
			ws->has_contact = wh->has_contact; ws->contact_body = wh->has_contact ? wh->contact_body : SGP_INVALID_ID;
			if (wh->has_contact) {
				vec_out(ws->contact_position, wh->contact_pos); vec_out(ws->contact_normal, wh->contact_normal);
				vec_out(ws->contact_longitudinal, wh->contact_long); vec_out(ws->contact_lateral, wh->contact_lat); vec_out(ws->contact_point_velocity, wh->contact_point_vel);
			}
			ws->suspension_lambda = wh->suspension.lambda + wh->max_up.lambda; ws->longitudinal_lambda = wh->longitudinal.lambda; ws->lateral_lambda = wh->lateral.lambda;
			ws->longitudinal_slip = wh->long_slip; ws->lateral_slip = wh->lat_slip;
		}
		s->engine_rpm = v->engine_rpm; s->current_gear = v->current_gear; s->clutch_friction = v->clutch_friction; s->active = v->active;
	}
	return SGP_OK;
}
SGO_API int sgo_vehicle_get_state(sgo_world* w, uint32_t id, sgp_vehicle_state* out) { return sgo_vehicle_get_states(w, id, 1, out); }

SGO_API int sgo_vehicle_enable_lean_controller(sgo_world* w, uint32_t id, int enabled)
{
	if (!vehicle_live(w, id)) return SGP_ERR_BAD_ID;
	w->vehicles[id].lean_enabled = w->vehicles[id].is_motorcycle && enabled;
	return SGP_OK;
}

SGO_API int sgo_vehicle_reset_drivetrain(sgo_world* w, uint32_t id, float rpm, float wheel_w)
{
	if (!vehicle_live(w, id)) return SGP_ERR_BAD_ID;
	sgo_vehicle* v = &w->vehicles[id];
	v->engine_rpm = rpm;
	for (int i = 0; i < v->num_wheels; ++i) v->wheels[i].angular_velocity = wheel_w;
	return SGP_OK;
}

static sgo_chassis chassis_load(const sgo_body* b)
{
	sgo_chassis c;
	c.pos = b->pos; c.rot = b->rot; c.v = b->linv; c.w = b->angv;
	c.im = b->inv_mass; c.inv_inertia_local = b->inv_inertia;
	c.I = world_inv_inertia(quat_to_m33(b->rot), b->inv_inertia);
	return c;
}

/* The dynamic bodies under the wheels of a vehicle as the rows see them: g[i] = NULL unless wheel i stands on a dynamic body that is awake for the
   solve (the effective inverse mass of a body still asleep is zero: before the wake-up of this step has happened -- the row setup -- the inverse
   mass of the body is what counts, see the caller); wheels that share a body share its state. */
static void vehicle_grounds_load(sgo_world* w, const sgo_vehicle* v, sgo_chassis* gs, sgo_chassis** g)
{
	for (int i = 0; i < v->num_wheels; ++i) {
		const sgo_wheel* wh = &v->wheels[i];
		g[i] = NULL;
		if (!wh->has_contact || !wh->ground_dynamic || !live(w, wh->contact_body)) continue;
		for (int j = 0; j < i; ++j) if (g[j] && v->wheels[j].contact_body == wh->contact_body) { g[i] = g[j]; break; }
		if (!g[i]) { gs[i] = chassis_load(&w->bodies[wh->contact_body]); g[i] = &gs[i]; }
	}
	for (int i = v->num_wheels; i < SGO_MAX_WHEELS; ++i) g[i] = NULL;
}

/* VehicleConstraint::OnStep for every vehicle whose chassis is awake, in two sweeps so that the result does not depend on the
   order of the vehicles: (A) all wheel casts (read-only on the bodies), (B) controller + row setup (writes the own chassis only).
   The cast visits every body (closest accepted hit; on equal distance the lower body id wins) -- the device walks the
   broad-phase grid instead and must find the same hit. */
static float cast_sphere_mesh(const sgo_body* M, v3 o, v3 d, float max_t, float rs, v3* n_out, v3* p_out);
static float cast_disc_mesh(const sgo_body* M, v3 o, v3 d, v3 e, v3 din, float disc_r, float rho, float max_t, v3* n_out, v3* p_out);

/* Slab test of a ray against a box, the broad-phase filter in front of every swept-sphere test (same arithmetic as the device's
 * ray_aabb; it is part of the result because the planes-only swept-sphere test of hulls and boxes is generous at their corners). */
static int ray_aabb(v3 o, v3 dir, v3 mn, v3 mx, float tmax)
{
	float t0 = 0.0f, t1 = tmax;
	const float oo[3] = { o.x, o.y, o.z }, dd[3] = { dir.x, dir.y, dir.z };
	const float lo[3] = { mn.x, mn.y, mn.z }, hi[3] = { mx.x, mx.y, mx.z };
	for (int a = 0; a < 3; ++a) {
		if (fabsf(dd[a]) < 1.0e-12f) { if (oo[a] < lo[a] - 1.0e-4f || oo[a] > hi[a] + 1.0e-4f) return 0; }
		else {
			float ta = (lo[a] - 1.0e-4f - oo[a]) / dd[a], tb = (hi[a] + 1.0e-4f - oo[a]) / dd[a];
			if (ta > tb) { const float tmp = ta; ta = tb; tb = tmp; }
			t0 = fmaxf(t0, ta); t1 = fminf(t1, tb);
			if (t0 > t1) return 0;
		}
	}
	return 1;
}
static int cast_reaches_bounds(const sgo_body* b, v3 o, v3 dir, float rs, float best)
{
	const float e = rs + 1.0e-3f;
	return ray_aabb(o, dir, V3(b->aabb_min.x - e, b->aabb_min.y - e, b->aabb_min.z - e), V3(b->aabb_max.x + e, b->aabb_max.y + e, b->aabb_max.z + e), best);
}

static void vehicles_pre_step(sgo_world* w, float dt)
{
	if (w->n_vehicles == 0) return;
	/* debugging aid: SGO_WHEEL_TRACE="<step>,<wheel>" prints the candidates of that wheel of vehicle 0 in that call of this function */
	static int dbg_calls = 0; int dbg_wheel = -1; const int dbg_step_now = ++dbg_calls;
	{ const char* e = getenv("SGO_WHEEL_TRACE"); if (e) { int st_ = 0, wh_ = 0; if (sscanf(e, "%d,%d", &st_, &wh_) == 2 && st_ == dbg_step_now) dbg_wheel = wh_; } }
	/* dense copy of the candidate bounds for the conservative reject (one cache-friendly stream instead of the body records) */
	float* bounds = (float*)malloc(sizeof(float) * 6 * (w->high ? w->high : 1));
	for (uint32_t j = 0; j < w->high; ++j) {
		const sgo_body* o = &w->bodies[j];
		const int cand = o->alive && !o->is_alias && !o->is_sensor && (o->layer == SGP_LAYER_NON_MOVING || o->layer == SGP_LAYER_MOVING);   /* tester object layer MOVING, CarPhysics.cpp:62 */
		float* bb = &bounds[6 * j];
		if (cand) { bb[0] = o->aabb_min.x; bb[1] = o->aabb_min.y; bb[2] = o->aabb_min.z; bb[3] = o->aabb_max.x; bb[4] = o->aabb_max.y; bb[5] = o->aabb_max.z; }
		else { bb[0] = bb[1] = bb[2] = 1.0f; bb[3] = bb[4] = bb[5] = -1.0f; }                          /* empty box: never overlaps */
	}
	#pragma omp parallel for schedule(dynamic, 4) if (g_threads > 1)
	for (uint32_t k = 0; k < w->n_vehicles; ++k) {
		sgo_vehicle* v = &w->vehicles[k];
		if (!v->alive) continue;
		v->active = live(w, v->body) && body_movable(&w->bodies[v->body]);
		/* A vehicle whose chassis sleeps still casts its wheels (VehicleConstraint::OnStep runs every step): the constraint is active when the chassis
		   OR a body a wheel touches is active, and an active vehicle constraint activates its chassis when the islands are built
		   (VehicleConstraint::BuildIslands -- UNVERIFIED: upstream).  So a ball rolling under a wheel of a parked car wakes the car although it never
		   touches the chassis.  The sleeping vehicle casts on a copy of its record: it stays frozen unless this step wakes it. */
		const int dormant = !v->active && live(w, v->body) && w->bodies[v->body].motion == SGP_MOTION_DYNAMIC && !w->bodies[v->body].active;
		if (!v->active && !dormant) continue;
		sgo_vehicle saved; if (dormant) saved = *v;
		int touched_active = 0;
		const sgo_chassis c = chassis_load(&w->bodies[v->body]);
		sgo_vehicle_pre_a(v, &c, dt);
		for (int i = 0; i < v->num_wheels; ++i) {
			sgo_wheel* wh = &v->wheels[i];
			float best = wh->cast_len; uint32_t bid = SGP_INVALID_ID; v3 bn = V3(0, 0, 0), bp = V3(0, 0, 0);
			const v3 e = v3_add(wh->cast_origin, v3_scale(wh->cast_dir, wh->cast_len));
			/* cheap reject only; it must never be stricter than the slab test below (which allows 1e-4 of slack and is the filter the device
			   applies), so it gets a wider margin -- a wheel 8e-5 m beside a hull's bounds once made the two disagree (tools/fuzz_parity.py) */
			const int cyl = v->tester == SGP_VEHICLE_TESTER_CYLINDER;
			const float reach = cyl ? wh->radius : v->cast_radius;      /* what the moving shape reaches around the path of its centre */
			const float m = reach + 2.0e-3f;
			const v3 lo = v3_sub(v3_min(wh->cast_origin, e), V3(m, m, m)), hi = v3_add(v3_max(wh->cast_origin, e), V3(m, m, m));
			for (uint32_t j = 0; j < w->high; ++j) {
				const float* bb = &bounds[6 * j];
				if (dbg_wheel >= 0 && i == dbg_wheel && getenv("SGO_WHEEL_TRACE_BODY") && (int)j == atoi(getenv("SGO_WHEEL_TRACE_BODY"))) fprintf(stderr, "[sgo wheel trace] body %u bounds %g %g %g .. %g %g %g | segment box %g %g %g .. %g %g %g | alive %d sensor %d layer %d alias %d shape %d\n", j, bb[0], bb[1], bb[2], bb[3], bb[4], bb[5], lo.x, lo.y, lo.z, hi.x, hi.y, hi.z, w->bodies[j].alive, w->bodies[j].is_sensor, w->bodies[j].layer, w->bodies[j].is_alias, w->bodies[j].shape_type);
				if (bb[3] < lo.x || bb[0] > hi.x || bb[4] < lo.y || bb[1] > hi.y || bb[5] < lo.z || bb[2] > hi.z) continue;
				if (j == v->body) continue;
				const sgo_body* o = &w->bodies[j];
				if (dbg_wheel >= 0 && i == dbg_wheel) fprintf(stderr, "[sgo wheel trace] step %d body %u: in segment box; reaches bounds %d; origin %g %g %g dir %g %g %g len %g aabb %g %g %g .. %g %g %g\n", dbg_step_now, j, cast_reaches_bounds(o, wh->cast_origin, wh->cast_dir, v->cast_radius, wh->cast_len), wh->cast_origin.x, wh->cast_origin.y, wh->cast_origin.z, wh->cast_dir.x, wh->cast_dir.y, wh->cast_dir.z, wh->cast_len, o->aabb_min.x, o->aabb_min.y, o->aabb_min.z, o->aabb_max.x, o->aabb_max.y, o->aabb_max.z);
				if (!cast_reaches_bounds(o, wh->cast_origin, wh->cast_dir, reach, wh->cast_len)) continue;      /* full length, not `best`: the answer must not depend on the visiting order */
				v3 n, p;
				float t;
				if (cyl) {
					/* the wheel itself (sgo_cast_disc): always over the whole travel; a hit behind the best one so far loses below */
					t = o->shape_type == SGP_SHAPE_MESH ? cast_disc_mesh(o, wh->cast_origin, wh->cast_dir, wh->cast_e, wh->cast_din, wh->disc_r, wh->cast_rho, wh->cast_len, &n, &p)
					                                    : sgo_cast_disc_body(o->shape_type, o->shape, o->hull, o->pos, quat_to_m33(o->rot), wh->cast_origin, wh->cast_dir, wh->cast_e, wh->cast_din, wh->disc_r, wh->cast_rho, wh->cast_len, &n, &p);
					if (t > best) t = -1.0f;
				} else
				t = o->shape_type == SGP_SHAPE_MESH ? cast_sphere_mesh(o, wh->cast_origin, wh->cast_dir, best, v->cast_radius, &n, &p)
				                                    : sgo_cast_sphere_body(o->shape_type, o->shape, o->hull, o->pos, quat_to_m33(o->rot), wh->cast_origin, wh->cast_dir, best, v->cast_radius, &n, &p);
				if (dbg_wheel >= 0 && i == dbg_wheel) fprintf(stderr, "[sgo wheel trace]   body %u: t %g n %g %g %g (best so far %g)\n", j, t, n.x, n.y, n.z, best);
				if (t < 0.0f || (!cyl && n.z < v->cos_max_slope)) continue;      /* (VehicleCollisionTesterCastCylinder has no slope limit -- UNVERIFIED: upstream) */
				if (t < best || bid == SGP_INVALID_ID) { best = t; bid = j; bn = n; bp = p; }
			}
			if (bid != SGP_INVALID_ID) {
				const sgo_body* o = &w->bodies[bid];
				const v3 gv = o->motion == SGP_MOTION_STATIC ? V3(0, 0, 0) : v3_add(o->linv, v3_cross(o->angv, v3_sub(bp, o->pos)));
				sgo_vehicle_set_hit(v, i, bid, best, bn, bp, gv, o->friction);
				wh->ground_dynamic = o->motion == SGP_MOTION_DYNAMIC;      /* the rows then act on it too: VehicleConstraint::SetupVelocityConstraint(.., *w->mContactBody, ..) */
				if (o->motion != SGP_MOTION_STATIC && o->active) touched_active = 1;
			}
		}
		if (dormant) {
			if (touched_active) { v->active = 1; w->bodies[v->body].can_sleep = -1; }      /* woken like a body an active one touches: after this step's collision detection */
			else *v = saved;
		}
	}
	free(bounds);
	/* a sleeping dynamic body under a wheel of an active vehicle wakes up (VehicleConstraint::BuildIslands activates the bodies the wheels touch);
	   like a body touched by an active one it is activated after this step's collision detection (find_contacts), so it gets no gravity this step */
	for (uint32_t k = 0; k < w->n_vehicles; ++k) {
		sgo_vehicle* v = &w->vehicles[k];
		if (!v->alive || !v->active) continue;
		for (int i = 0; i < v->num_wheels; ++i) {
			const sgo_wheel* wh = &v->wheels[i];
			if (wh->has_contact && wh->ground_dynamic && !w->bodies[wh->contact_body].active) w->bodies[wh->contact_body].can_sleep = -1;
		}
	}
	for (uint32_t k = 0; k < w->n_vehicles; ++k) {
		sgo_vehicle* v = &w->vehicles[k];
		if (!v->alive || !v->active) continue;
		sgo_body* b = &w->bodies[v->body];
		sgo_chassis c = chassis_load(b);
		sgo_chassis gs[SGO_MAX_WHEELS]; sgo_chassis* g[SGO_MAX_WHEELS];
		vehicle_grounds_load(w, v, gs, g);
		if (sgo_vehicle_pre_b(v, &c, g, dt)) b->sleep_timer = 0.0f;
		b->linv = c.v; b->angv = c.w;
	}
}

/* mode 0 warm start, 1 velocity iteration, 2 position iteration.  The vehicles in index order (the constraints of an island in their order, in
   Jolt): two vehicles with a wheel on the same dynamic body, or one standing on the other, see each other's impulses in that order. */
static void vehicles_solve(sgo_world* w, int mode, float dt)
{
	for (uint32_t k = 0; k < w->n_vehicles; ++k) {
		sgo_vehicle* v = &w->vehicles[k];
		if (!v->alive || !v->active) continue;
		sgo_body* b = &w->bodies[v->body];
		sgo_chassis c = chassis_load(b);
		sgo_chassis gs[SGO_MAX_WHEELS]; sgo_chassis* g[SGO_MAX_WHEELS];
		vehicle_grounds_load(w, v, gs, g);
		if (mode == 0) sgo_vehicle_warm_start(v, &c, g);
		else if (mode == 1) sgo_vehicle_solve_velocity(v, &c, g, dt);
		else sgo_vehicle_solve_position(v, &c, g, w->st.baumgarte);
		if (mode == 2) { b->pos = c.pos; b->rot = c.rot; } else { b->linv = c.v; b->angv = c.w; }
		for (int i = 0; i < v->num_wheels; ++i) {
			if (!g[i] || g[i] != &gs[i]) continue;                      /* (a body two wheels share is written once, by the first of them) */
			sgo_body* o = &w->bodies[v->wheels[i].contact_body];
			if (mode == 2) { o->pos = gs[i].pos; o->rot = gs[i].rot; } else { o->linv = gs[i].v; o->angv = gs[i].w; }
		}
	}
}

static int cmp_prev(const void* a, const void* b)
{
	const keyidx* x = (const keyidx*)a; const keyidx* y = (const keyidx*)b;
	return (x->key > y->key) - (x->key < y->key);
}

/* Debugging aid: with SGO_NAN_TRACE=1 in the environment, reports the first stage of a step after which some body's state is not finite. */
static void nan_trace(sgo_world* w, const char* stage)
{
	static int enabled = -1, reported = 0;
	if (enabled < 0) { const char* e = getenv("SGO_NAN_TRACE"); enabled = (e && e[0] == '1') ? 1 : 0; }
	if (!enabled || reported) return;
	for (uint32_t k = 0; (stage[0] == '4' || stage[0] == '5') && k < w->n_cons; ++k) {
		const sgo_constraint* c = &w->cons[k];
		float s = c->n.x + c->n.y + c->n.z + c->t1.x + c->t1.y + c->t1.z + c->friction;
		for (int i = 0; i < c->np; ++i) s += c->pt[i].r1.x + c->pt[i].r1.y + c->pt[i].r1.z + c->pt[i].r2.x + c->pt[i].r2.y + c->pt[i].r2.z + c->pt[i].bias + c->pt[i].eff_n + c->pt[i].eff_t1 + c->pt[i].eff_t2 + c->pt[i].lam_n + c->pt[i].lam_t1 + c->pt[i].lam_t2;
		if (!(s - s == 0.0f)) {
			fprintf(stderr, "[sgo nan trace] constraint (%u shape %d, %u shape %d) not finite after stage '%s': n %g %g %g t1 %g %g %g np %d", c->a, w->bodies[c->a].shape_type, c->b, w->bodies[c->b].shape_type, stage, c->n.x, c->n.y, c->n.z, c->t1.x, c->t1.y, c->t1.z, c->np);
			for (int i = 0; i < c->np; ++i) fprintf(stderr, " | r1 %g %g %g bias %g eff %g %g %g lam %g %g %g", c->pt[i].r1.x, c->pt[i].r1.y, c->pt[i].r1.z, c->pt[i].bias, c->pt[i].eff_n, c->pt[i].eff_t1, c->pt[i].eff_t2, c->pt[i].lam_n, c->pt[i].lam_t1, c->pt[i].lam_t2);
			fprintf(stderr, "\n"); reported = 1; return;
		}
	}
	for (uint32_t i = 0; i < w->high; ++i) {
		const sgo_body* b = &w->bodies[i];
		if (!b->alive) continue;
		const float s = b->pos.x + b->pos.y + b->pos.z + b->linv.x + b->linv.y + b->linv.z + b->angv.x + b->angv.y + b->angv.z + b->rot.x + b->rot.y + b->rot.z + b->rot.w;
		if (!(s - s == 0.0f)) { fprintf(stderr, "[sgo nan trace] body %u not finite after stage '%s' (linv %g %g %g angv %g %g %g)\n", i, stage, b->linv.x, b->linv.y, b->linv.z, b->angv.x, b->angv.y, b->angv.z); reported = 1; return; }
	}
}

SGO_API int sgo_world_step(sgo_world* w, float dt)
{
	if (!w || !(dt > 0.0f)) return SGP_ERR_INVALID;
	memset(&w->stats, 0, sizeof(w->stats));

	/* 0. step listeners: VehicleConstraint::OnStep (wheel casts, controller, row setup) */
	nan_trace(w, "edits before the step");
	vehicles_pre_step(w, dt);
	nan_trace(w, "0 vehicle pre-step");

	/* 1. MotionProperties::ApplyForceTorqueAndDragInternal (JobApplyGravity) */
	#pragma omp parallel for schedule(static, 1024) if (g_threads > 1)
	for (uint32_t i = 0; i < w->high; ++i) {
		sgo_body* b = &w->bodies[i];
		b->linv_pre = b->linv;
		if (!b->alive || !body_movable(b)) continue;
		b->linv = v3_add(b->linv, v3_scale(v3_add(v3_scale(w->gravity, b->gravity_factor), v3_scale(b->force, b->inv_mass)), dt));
		const sym33 Iw = world_inv_inertia(quat_to_m33(b->rot), b->inv_inertia);
		b->angv = v3_add(b->angv, v3_scale(sym33_mul(Iw, b->torque), dt));
		b->linv = v3_scale(b->linv, fmaxf(0.0f, 1.0f - b->lin_damp * dt));
		b->angv = v3_scale(b->angv, fmaxf(0.0f, 1.0f - b->ang_damp * dt));
		const float l2 = v3_len_sq(b->linv), ml = w->st.max_linear_velocity;
		if (l2 > ml * ml) b->linv = v3_scale(b->linv, ml / sqrtf(l2));
		const float a2 = v3_len_sq(b->angv), ma = w->st.max_angular_velocity;
		if (a2 > ma * ma) b->angv = v3_scale(b->angv, ma / sqrtf(a2));
		b->force = V3(0, 0, 0); b->torque = V3(0, 0, 0);
	}

	/* 2-3. broad phase, narrow phase, contact constraints */
	broad_phase(w);
	find_contacts(w, dt);
	/* is anybody awake in this step (kinematic bodies count; what the contacts above woke is awake by now)?  A step nobody is awake in is the identity -- and
	   leaves the contact cache as it is (step 10) */
	int any_awake = 0;
	for (uint32_t i = 0; i < w->high; ++i) {
		sgo_body* b = &w->bodies[i];
		b->awake_step = b->alive && !b->is_alias && body_is_active_for_pairs(b);
		if (b->awake_step) any_awake = 1;
	}

	nan_trace(w, "1-3 forces, collision");
	/* 4. colouring and solve order */
	colour_constraints(w);
	w->order = (uint32_t*)realloc(w->order, sizeof(uint32_t) * (w->n_cons ? w->n_cons : 1));
	for (uint32_t k = 0; k < w->n_cons; ++k) w->order[k] = k;
	g_sort_world = w;
	qsort(w->order, w->n_cons, sizeof(uint32_t), cmp_order);
	int ncol = 0, nreg = 0; uint32_t novf = 0;
	for (uint32_t k = 0; k < w->n_cons; ++k) {
		if (w->cons[k].colour + 1 > ncol) ncol = w->cons[k].colour + 1;
		if (w->cons[k].colour == SGO_OVERFLOW_COLOUR) ++novf; else if (w->cons[k].colour + 1 > nreg) nreg = w->cons[k].colour + 1;
	}

	/* colour segments of the solve order: constraints of one colour (except the overflow colour) share no movable body, so a
	   segment may be walked in any order or in parallel without changing a single bit */
	uint32_t seg[SGO_MAX_COLOURS + 1];
	{
		uint32_t k = 0;
		for (int c = 0; c < SGO_MAX_COLOURS; ++c) { seg[c] = k; while (k < w->n_cons && w->cons[w->order[k]].colour == c) ++k; }
		seg[SGO_MAX_COLOURS] = k;
	}
	#define SOLVE_PASS(fn) do { \
		for (int c_ = 0; c_ < SGO_MAX_COLOURS; ++c_) { \
			const uint32_t b_ = seg[c_], e_ = seg[c_ + 1]; \
			if (c_ != SGO_OVERFLOW_COLOUR && g_threads > 1 && e_ - b_ > 64) { \
				_Pragma("omp parallel for schedule(static, 64)") \
				for (uint32_t k_ = b_; k_ < e_; ++k_) fn(w, &w->cons[w->order[k_]]); \
			} else for (uint32_t k_ = b_; k_ < e_; ++k_) fn(w, &w->cons[w->order[k_]]); \
		} } while (0)

	nan_trace(w, "4 colouring, setup");
	/* 5. warm start + velocity iterations */
	/* (non-contact constraints -- here: the vehicles -- go first in every pass, as in PhysicsSystem::JobSolveVelocityConstraints) */
	if (w->st.warm_start) { vehicles_solve(w, 0, dt); SOLVE_PASS(warm_start_constraint); }
	for (int it = 0; it < w->st.num_velocity_steps; ++it) { vehicles_solve(w, 1, dt); SOLVE_PASS(solve_velocity_constraint); }

	nan_trace(w, "5 warm start + velocity iterations");
	/* 6. integrate positions (Body::AddPositionStep / AddRotationStep) */
	#pragma omp parallel for schedule(static, 1024) if (g_threads > 1)
	for (uint32_t i = 0; i < w->high; ++i) {
		sgo_body* b = &w->bodies[i];
		if (!b->alive || !b->active || b->motion == SGP_MOTION_STATIC) continue;
		if (b->motion == SGP_MOTION_DYNAMIC) {
			const float l2 = v3_len_sq(b->linv), ml = w->st.max_linear_velocity;
			if (l2 > ml * ml) b->linv = v3_scale(b->linv, ml / sqrtf(l2));
			const float a2 = v3_len_sq(b->angv), ma = w->st.max_angular_velocity;
			if (a2 > ma * ma) b->angv = v3_scale(b->angv, ma / sqrtf(a2));
		}
		b->pos = v3_add(b->pos, v3_scale(b->linv, dt));
		b->rot = quat_add_rotation_step(b->rot, v3_scale(b->angv, dt));
		if (b->shape_type == SGP_SHAPE_MESH) sync_mesh_aliases(w, i);      /* (a kinematic mesh body: only its own three slots are touched) */
	}

	nan_trace(w, "6 integrate");
	/* 7. position iterations */
	for (int it = 0; it < w->st.num_position_steps; ++it) { vehicles_solve(w, 2, dt); SOLVE_PASS(solve_position_constraint); }

	nan_trace(w, "7 position iterations");
	/* 8. bounds, sleeping */
	#pragma omp parallel for schedule(static, 1024) if (g_threads > 1)
	for (uint32_t i = 0; i < w->high; ++i) { sgo_body* b = &w->bodies[i]; if (b->alive && b->active) body_update_aabb(b); }
	update_sleeping(w, dt);

	for (uint32_t i = 0; i < w->high; ++i) { w->bodies[i].fresh = w->bodies[i].cache_invalid; w->bodies[i].cache_invalid = 0; }      /* every live body has now been through a step with its present shape */

	nan_trace(w, "8 bounds, sleeping");
	/* 9. buoyancy (Substrata's own sweep after Update) */
	if (w->water_enabled) buoyancy_sweep(w, dt);

	nan_trace(w, "9 buoyancy");
	/* 10. contact cache for the next step.  A step without an awake body keeps the cache it found: the contacts of a pile that fell asleep as a whole are
	   there when it wakes (warm start, colours, the body-pair cache's manifolds) -- ContactConstraintManager keeps the cached contacts of sleeping bodies
	   (UNVERIFIED: upstream).  Only the whole-world case: a pair that sleeps while others are awake loses its entry with the next step (docs/GAPS.md). */
	const uint32_t n_cons_step = w->n_cons;
	uint32_t n_reused_step = 0;
	for (uint32_t k = 0; k < w->n_cons; ++k) n_reused_step += (uint32_t)w->cons[k].reused;
	if (any_awake) {
		/* the contacts of sleeping pairs stay in the cache (ContactConstraintManager keeps the cached contacts of bodies that are not active -- UNVERIFIED:
		   upstream): an entry of the cache this step found is carried over when neither of its bodies was awake in this step (such a pair made no
		   constraint of its own: no duplicate) and neither was created or reshaped since the last step; it is looked up like any other when the pair
		   wakes -- warm start, the body-pair cache's manifold, 'persisted' -- and is never solved, counted or reported */
		for (uint32_t k = 0; k < w->n_prev; ++k) {
			const sgo_constraint* e = &w->prev[k];
			const sgo_body* A = &w->bodies[e->a]; const sgo_body* B = &w->bodies[e->b];
			if (!A->alive || !B->alive || A->awake_step || B->awake_step || A->fresh || B->fresh) continue;
			if (w->n_cons == w->cap_cons) { w->cap_cons = w->cap_cons + w->cap_cons / 2 + 1024; w->cons = (sgo_constraint*)realloc(w->cons, sizeof(sgo_constraint) * w->cap_cons); }
			w->cons[w->n_cons] = *e; w->cons[w->n_cons].carried = 1;
			w->n_cons++;
		}
		sgo_constraint* t = w->prev; w->prev = w->cons; w->cons = t;
		const uint32_t tc = w->cap_prev; w->cap_prev = w->cap_cons; w->cap_cons = tc;
		w->n_prev = w->n_cons;
		keyidx* ki = (keyidx*)malloc(sizeof(keyidx) * (w->n_prev ? w->n_prev : 1));
		for (uint32_t k = 0; k < w->n_prev; ++k) { ki[k].key = w->prev[k].key; ki[k].idx = k; }
		qsort(ki, w->n_prev, sizeof(keyidx), cmp_prev);
		w->prev_keys_sorted = (uint64_t*)realloc(w->prev_keys_sorted, sizeof(uint64_t) * (w->n_prev ? w->n_prev : 1));
		w->prev_idx_sorted = (uint32_t*)realloc(w->prev_idx_sorted, sizeof(uint32_t) * (w->n_prev ? w->n_prev : 1));
		for (uint32_t k = 0; k < w->n_prev; ++k) { w->prev_keys_sorted[k] = ki[k].key; w->prev_idx_sorted[k] = ki[k].idx; }
		free(ki);
	}

	/* stats */
	w->stats.num_bodies = w->n_alive;
	for (uint32_t i = 0; i < w->high; ++i) {
		const sgo_body* b = &w->bodies[i];
		if (!b->alive || b->is_alias) continue;
		if (b->active) w->stats.num_active++;
		if (b->layer >= 0 && b->layer < SGP_NUM_LAYERS) w->stats.layer_counts[b->layer]++;
	}
	w->stats.num_pairs = w->n_pairs;
	w->stats.num_manifolds = n_cons_step;
	w->stats.num_colours = (uint32_t)nreg;      /* regular colours only, like the device's colour table; the overflow colour is num_overflow_constraints */
	w->stats.num_overflow_constraints = novf;
	w->stats.num_cached_manifolds = n_reused_step;
	/* vehicles that share a movable body with a vehicle of lower index (the device defers their rows; here they are simply later in the loop) */
	if (w->n_vehicles) {
		uint32_t* first = (uint32_t*)malloc(sizeof(uint32_t) * (w->high ? w->high : 1));
		for (uint32_t i = 0; i < w->high; ++i) first[i] = 0xFFFFFFFFu;
		for (int pass = 0; pass < 2; ++pass) for (uint32_t k = 0; k < w->n_vehicles; ++k) {
			const sgo_vehicle* v = &w->vehicles[k];
			if (!v->alive || !v->active || v->body >= w->high) continue;
			int lost = 0;
			if (pass == 0) { if (k < first[v->body]) first[v->body] = k; } else if (first[v->body] != k) lost = 1;
			for (int i = 0; i < v->num_wheels; ++i) {
				const sgo_wheel* wh = &v->wheels[i];
				if (!wh->has_contact || !wh->ground_dynamic || wh->contact_body >= w->high) continue;
				if (pass == 0) { if (k < first[wh->contact_body]) first[wh->contact_body] = k; } else if (first[wh->contact_body] != k) lost = 1;
			}
			w->stats.num_deferred_vehicles += (uint32_t)lost;
		}
		free(first);
	}
	/* activation events raised since the end of the previous step (edits between steps included) */
	w->stats.num_activated = (uint32_t)(w->tot_act - w->rep_act); w->rep_act = w->tot_act;
	w->stats.num_deactivated = (uint32_t)(w->tot_deact - w->rep_deact); w->rep_deact = w->tot_deact;
	return SGP_OK;
}

SGO_API int sgo_world_step_n(sgo_world* w, float dt, uint32_t n)
{
	for (uint32_t i = 0; i < n; ++i) { const int r = sgo_world_step(w, dt); if (r != SGP_OK) return r; }
	return SGP_OK;
}

SGO_API int sgo_world_stats(sgo_world* w, sgp_step_stats* out) { *out = w->stats; return SGP_OK; }

static int cmp_body_event(const void* a, const void* b)
{
	const uint32_t x = ((const sgp_body_event*)a)->id, y = ((const sgp_body_event*)b)->id;
	return (x > y) - (x < y);
}
static int cmp_contact_event(const void* a, const void* b)
{
	const sgp_contact_event* x = (const sgp_contact_event*)a; const sgp_contact_event* y = (const sgp_contact_event*)b;
	if (x->id1 != y->id1) return (x->id1 > y->id1) - (x->id1 < y->id1);
	if (x->id2 != y->id2) return (x->id2 > y->id2) - (x->id2 < y->id2);
	/* several events of one pair (the manifolds of a body against a mesh, or several steps drained together): a total order, so that
	   the result does not depend on the sort or on the arrival order */
	for (int k = 0; k < 3; ++k) if (x->base_offset[k] != y->base_offset[k]) return (x->base_offset[k] > y->base_offset[k]) - (x->base_offset[k] < y->base_offset[k]);
	for (int k = 0; k < 3; ++k) if (x->normal[k] != y->normal[k]) return (x->normal[k] > y->normal[k]) - (x->normal[k] < y->normal[k]);
	return (x->penetration > y->penetration) - (x->penetration < y->penetration);
}

SGO_API int sgo_world_drain_events(sgo_world* w, int kind, void* out, uint32_t cap, uint32_t* n_out)
{
	if (!w || !n_out) return SGP_ERR_INVALID;
	if (kind <= SGP_EVENT_ENTERED_WATER) {
		sgp_body_event* src; uint32_t* n;
		if (kind == SGP_EVENT_ACTIVATED) { src = w->ev_act; n = &w->n_act; }
		else if (kind == SGP_EVENT_DEACTIVATED) { src = w->ev_deact; n = &w->n_deact; }
		else { src = w->ev_water; n = &w->n_water; }
		qsort(src, *n, sizeof(sgp_body_event), cmp_body_event);
		const uint32_t m = *n < cap ? *n : cap;
		if (out && m) memcpy(out, src, sizeof(sgp_body_event) * m);
		*n_out = *n; *n = 0;
	} else {
		sgp_contact_event* src; uint32_t* n;
		if (kind == SGP_EVENT_CONTACT_ADDED) { src = w->ev_added; n = &w->n_added; } else { src = w->ev_pers; n = &w->n_pers; }
		qsort(src, *n, sizeof(sgp_contact_event), cmp_contact_event);
		const uint32_t m = *n < cap ? *n : cap;
		if (out && m) memcpy(out, src, sizeof(sgp_contact_event) * m);
		*n_out = *n; *n = 0;
	}
	return SGP_OK;
}

/* ------------------------------------------------------------------------------------------------ */
/* multi-GPU tiles: the same export / import semantics as the device library, so that the tile exchange  */
/* (substrata_amd/tiles.py) can be exercised on CPU with gloo                                                            */

static int cmp_ghost_rec(const void* a, const void* b)
{
	const uint64_t x = ((const sgp_ghost_record*)a)->global_id, y = ((const sgp_ghost_record*)b)->global_id;
	return (x > y) - (x < y);
}

SGO_API int sgo_world_export_boundary(sgo_world* w, const float lo[3], const float hi[3], float margin, sgp_ghost_record* out, uint32_t cap, uint32_t* n_out)
{
	uint32_t n = 0;
	const float large_r = w->desc.large_body_radius;
	for (uint32_t i = 0; i < w->high; ++i) {
		const sgo_body* b = &w->bodies[i];
		if (!b->alive || w->is_ghost[i] || b->motion == SGP_MOTION_STATIC) continue;
		if (body_bounding_radius(b) > large_r) continue;
		const int crosses = b->aabb_min.x - margin < lo[0] || b->aabb_min.y - margin < lo[1] || b->aabb_min.z - margin < lo[2] ||
		                    b->aabb_max.x + margin >= hi[0] || b->aabb_max.y + margin >= hi[1] || b->aabb_max.z + margin >= hi[2];
		if (!crosses) continue;
		if (n < cap) {
			sgp_ghost_record* r = &out[n];
			memset(r, 0, sizeof(*r));
			r->pos[0] = b->pos.x; r->pos[1] = b->pos.y; r->pos[2] = b->pos.z;
			r->rot[0] = b->rot.x; r->rot[1] = b->rot.y; r->rot[2] = b->rot.z; r->rot[3] = b->rot.w;
			r->lin_vel[0] = b->linv.x; r->lin_vel[1] = b->linv.y; r->lin_vel[2] = b->linv.z;
			r->ang_vel[0] = b->angv.x; r->ang_vel[1] = b->angv.y; r->ang_vel[2] = b->angv.z;
			r->shape_type = b->shape_type; memcpy(r->shape, b->shape, sizeof(r->shape)); r->shape[3] = 0.0f;
			r->mass = b->mass; r->friction = b->friction; r->restitution = b->restitution;
			r->motion_type = (uint32_t)b->motion; r->global_id = i;
			r->userdata = b->userdata; r->gravity_factor = b->gravity_factor; r->linear_damping = b->lin_damp; r->angular_damping = b->ang_damp;
			r->flags = ((uint32_t)b->layer & SGP_GHOST_FLAG_LAYER_MASK) | (b->is_sensor ? SGP_GHOST_FLAG_SENSOR : 0u) | (b->allow_sleep ? SGP_GHOST_FLAG_ALLOW_SLEEP : 0u) | (b->zero_lin_drag ? SGP_GHOST_FLAG_ZERO_DRAG : 0u);
			for (uint32_t v = 0; v < w->n_vehicles; ++v) if (w->vehicles[v].alive && w->vehicles[v].body == i) r->flags |= SGP_GHOST_FLAG_CHASSIS;
		}
		++n;
	}
	*n_out = n;
	qsort(out, n < cap ? n : cap, sizeof(sgp_ghost_record), cmp_ghost_rec);
	return SGP_OK;
}

SGO_API int sgo_world_import_ghosts(sgo_world* w, const sgp_ghost_record* in, uint32_t n)
{
	uint64_t* ngid = (uint64_t*)malloc(sizeof(uint64_t) * (n ? n : 1));
	uint32_t* nlid = (uint32_t*)malloc(sizeof(uint32_t) * (n ? n : 1));
	int* kept = (int*)calloc(w->n_ghosts ? w->n_ghosts : 1, sizeof(int));
	uint32_t nn = 0;
	for (uint32_t k = 0; k < n; ++k) {
		int found = -1;
		for (uint32_t g = 0; g < w->n_ghosts; ++g) if (w->ghost_gid[g] == in[k].global_id && !kept[g] && live(w, w->ghost_lid[g])) { found = (int)g; break; }
		if (found >= 0) {
			const uint32_t id = w->ghost_lid[found];
			kept[found] = 1;
			sgo_body_set_pose_vel(w, id, in[k].pos, in[k].rot, in[k].lin_vel, in[k].ang_vel);
			sgo_body_activate(w, id);
			ngid[nn] = in[k].global_id; nlid[nn] = id; ++nn;
			continue;
		}
		sgp_body_desc d; sgo_default_body_desc(&d);
		memcpy(d.pos, in[k].pos, 12); memcpy(d.rot, in[k].rot, 16); memcpy(d.lin_vel, in[k].lin_vel, 12); memcpy(d.ang_vel, in[k].ang_vel, 12);
		d.shape_type = in[k].shape_type; memcpy(d.shape, in[k].shape, 16);
		d.motion_type = SGP_MOTION_KINEMATIC;
		/* layer and sensor flag of the original body (a sensor near the border is a sensor next door too); a kinematic ghost lives on a moving layer */
		d.layer = (int32_t)(in[k].flags & SGP_GHOST_FLAG_LAYER_MASK);
		if (d.layer == SGP_LAYER_NON_MOVING) d.layer = SGP_LAYER_MOVING;
		if (d.layer == SGP_LAYER_NON_MOVING_NON_COLLIDABLE) d.layer = SGP_LAYER_MOVING_NON_COLLIDABLE;
		d.is_sensor = (in[k].flags & SGP_GHOST_FLAG_SENSOR) ? 1 : 0;
		d.mass = in[k].mass; d.friction = in[k].friction; d.restitution = in[k].restitution;
		d.activate = 1; d.userdata = in[k].userdata;       /* a ray or an event that meets the ghost names the object, like its owner would */
		uint32_t id = SGP_INVALID_ID;
		const int r = sgo_body_add(w, &d, &id);
		if (r == SGP_OK) { w->is_ghost[id] = 1; ngid[nn] = in[k].global_id; nlid[nn] = id; ++nn; }
		else if (r != SGP_ERR_REJECTED) { free(ngid); free(nlid); free(kept); return r; }
	}
	/* remove the ghosts that left the set, ascending local id */
	for (uint32_t id = 0; id < w->high; ++id) {
		int gone = 0;
		for (uint32_t g = 0; g < w->n_ghosts; ++g) if (w->ghost_lid[g] == id && !kept[g] && live(w, id) && w->is_ghost[id]) { gone = 1; break; }
		if (gone) { w->is_ghost[id] = 0; sgo_body_remove(w, id); }
	}
	free(w->ghost_gid); free(w->ghost_lid); free(kept);
	w->ghost_gid = ngid; w->ghost_lid = nlid; w->n_ghosts = nn;
	return SGP_OK;
}

/* ------------------------------------------------------------------------------------------------ */
/* direct access for unit tests                                                                      */

static sgo_shape shape_from_desc(const sgp_body_desc* d);
/* Narrow phase on two body descs of THIS world (hull ids resolve against its hull table). */
SGO_API int sgo_world_collide_pair(sgo_world* w, const sgp_body_desc* a, const sgp_body_desc* b, float max_sep, float* normal, int* np, float* p1, float* p2)
{
	sgo_shape sa = shape_from_desc(a), sb = shape_from_desc(b);
	const sgp_body_desc* dd[2] = { a, b }; sgo_shape* ss[2] = { &sa, &sb };
	for (int k = 0; k < 2; ++k) {
		if (dd[k]->shape_type == SGP_SHAPE_HULL) { const uint32_t hid = (uint32_t)dd[k]->shape[0]; if (hid < 1 || hid >= w->n_hulls) return -1; ss[k]->hull = w->hulls[hid]; }
		else if (dd[k]->shape_type == SGP_SHAPE_BOX) ss[k]->hull = w->hulls[0];
	}
	sgo_manifold m;
	if (!sgo_collide(&sa, &sb, max_sep, &m)) { *np = 0; return 0; }
	normal[0] = m.n.x; normal[1] = m.n.y; normal[2] = m.n.z;
	*np = m.np;
	for (int i = 0; i < m.np; ++i) {
		p1[3 * i] = m.p1[i].x; p1[3 * i + 1] = m.p1[i].y; p1[3 * i + 2] = m.p1[i].z;
		p2[3 * i] = m.p2[i].x; p2[3 * i + 1] = m.p2[i].y; p2[3 * i + 2] = m.p2[i].z;
	}
	return 1;
}

static sgo_shape shape_from_desc(const sgp_body_desc* d)
{
	sgo_shape s;
	s.pos = V3(d->pos[0], d->pos[1], d->pos[2]);
	quat q = { d->rot[0], d->rot[1], d->rot[2], d->rot[3] };
	s.R = quat_to_m33(q);
	s.type = d->shape_type;
	memcpy(s.p, d->shape, sizeof(s.p));
	s.hull = NULL;
	return s;
}

/* out: normal[3], np, p1[4][3], p2[4][3].  Returns 1 on contact. */
SGO_API int sgo_collide_pair(const sgp_body_desc* a, const sgp_body_desc* b, float max_sep, float* normal, int* np, float* p1, float* p2)
{
	const sgo_shape sa = shape_from_desc(a), sb = shape_from_desc(b);
	sgo_manifold m;
	if (!sgo_collide(&sa, &sb, max_sep, &m)) { *np = 0; return 0; }
	normal[0] = m.n.x; normal[1] = m.n.y; normal[2] = m.n.z;
	*np = m.np;
	for (int i = 0; i < m.np; ++i) {
		p1[3 * i] = m.p1[i].x; p1[3 * i + 1] = m.p1[i].y; p1[3 * i + 2] = m.p1[i].z;
		p2[3 * i] = m.p2[i].x; p2[3 * i + 1] = m.p2[i].y; p2[3 * i + 2] = m.p2[i].z;
	}
	return 1;
}

/* MeshShapeSettings::Create */
SGO_API int sgo_mesh_create_with_materials(sgo_world* w, const float* verts, uint32_t nv, const uint32_t* idx, uint32_t nt, const uint32_t* mats, sgp_mesh_info* info);
SGO_API int sgo_mesh_create(sgo_world* w, const float* verts, uint32_t nv, const uint32_t* idx, uint32_t nt, sgp_mesh_info* info) { return sgo_mesh_create_with_materials(w, verts, nv, idx, nt, NULL, info); }
SGO_API int sgo_mesh_create_with_materials(sgo_world* w, const float* verts, uint32_t nv, const uint32_t* idx, uint32_t nt, const uint32_t* mats, sgp_mesh_info* info)
{
	if (!w || !verts || !idx || !info || nv < 3 || nt < 1) return SGP_ERR_INVALID;
	for (uint32_t k = 0; k < 3 * nt; ++k) if (idx[k] >= nv) return SGP_ERR_INVALID;
	for (uint32_t k = 0; k < 3 * nv; ++k) if (!isfinite(verts[k])) return SGP_ERR_INVALID;
	sgo_mesh* m = (sgo_mesh*)calloc(1, sizeof(sgo_mesh));
	m->nv = nv; m->nt = nt;
	m->verts = (v3*)malloc(sizeof(v3) * nv); m->tris = (uint32_t*)malloc(sizeof(uint32_t) * 3 * nt);
	memcpy(m->tris, idx, sizeof(uint32_t) * 3 * nt);
	m->mats = (uint32_t*)calloc(nt, sizeof(uint32_t));
	if (mats) memcpy(m->mats, mats, sizeof(uint32_t) * nt);
	m->edges = (unsigned char*)malloc(nt);
	sgo_mesh_active_edges(verts, idx, nt, (double)SGO_ACTIVE_EDGE_COS, m->edges);
	v3 mn = V3(3.4e38f, 3.4e38f, 3.4e38f), mx = V3(-3.4e38f, -3.4e38f, -3.4e38f); float br = 0.0f;
	for (uint32_t k = 0; k < nv; ++k) { const v3 p = V3(verts[3 * k], verts[3 * k + 1], verts[3 * k + 2]); m->verts[k] = p; mn = v3_min(mn, p); mx = v3_max(mx, p); br = fmaxf(br, v3_len(p)); }
	m->aabb_min = mn; m->aabb_max = mx; m->bound_radius = br;
	uint32_t id;
	if (w->n_free_mesh_ids) id = w->free_mesh_ids[--w->n_free_mesh_ids];
	else {
		if (w->n_meshes == w->cap_meshes) { w->cap_meshes *= 2; w->meshes = (sgo_mesh**)realloc(w->meshes, sizeof(sgo_mesh*) * w->cap_meshes); }
		id = w->n_meshes++;
	}
	w->meshes[id] = m;
	memset(info, 0, sizeof(*info));
	info->mesh_id = id; info->num_vertices = nv; info->num_triangles = nt; info->num_nodes = 0;
	info->aabb_min[0] = mn.x; info->aabb_min[1] = mn.y; info->aabb_min[2] = mn.z; info->aabb_max[0] = mx.x; info->aabb_max[1] = mx.y; info->aabb_max[2] = mx.z;
	return SGP_OK;
}

/* the active-edge bits of a mesh's triangles, in the caller's triangle order (bit k: edge k = v[k] - v[k + 1] collides with its own normal) */
SGO_API int sgo_mesh_edge_flags(sgo_world* w, uint32_t mesh_id, uint8_t* out, uint32_t cap)
{
	if (!w || !out || mesh_id < 1 || mesh_id >= w->n_meshes || !w->meshes[mesh_id]) return SGP_ERR_BAD_ID;
	const sgo_mesh* m = w->meshes[mesh_id];
	for (uint32_t t = 0; t < m->nt && t < cap; ++t) out[t] = m->edges[t];
	return SGP_OK;
}

static int shape_in_use(const sgo_world* w, int type, const void* p)
{
	for (uint32_t i = 0; i < w->high; ++i) { const sgo_body* b = &w->bodies[i]; if (b->alive && !b->is_alias && b->shape_type == type && (type == SGP_SHAPE_MESH ? (const void*)b->mesh : (const void*)b->hull) == p) return 1; }
	return 0;
}
SGO_API int sgo_mesh_destroy(sgo_world* w, uint32_t id)
{
	if (!w || id < 1 || id >= w->n_meshes || !w->meshes[id]) return SGP_ERR_BAD_ID;
	if (shape_in_use(w, SGP_SHAPE_MESH, w->meshes[id])) return SGP_ERR_REJECTED;
	free(w->meshes[id]->verts); free(w->meshes[id]->tris); free(w->meshes[id]->mats); free(w->meshes[id]->edges); free(w->meshes[id]); w->meshes[id] = NULL;
	w->free_mesh_ids = (uint32_t*)realloc(w->free_mesh_ids, sizeof(uint32_t) * (w->n_free_mesh_ids + 1));
	w->free_mesh_ids[w->n_free_mesh_ids++] = id;
	return SGP_OK;
}
SGO_API int sgo_hull_destroy(sgo_world* w, uint32_t id)
{
	if (!w || id < 1 || id >= w->n_hulls || !w->hulls[id]) return SGP_ERR_BAD_ID;
	if (shape_in_use(w, SGP_SHAPE_HULL, w->hulls[id])) return SGP_ERR_REJECTED;
	free(w->hulls[id]); w->hulls[id] = NULL;
	w->free_hull_ids = (uint32_t*)realloc(w->free_hull_ids, sizeof(uint32_t) * (w->n_free_hull_ids + 1));
	w->free_hull_ids[w->n_free_hull_ids++] = id;
	return SGP_OK;
}

/* Every triangle of mesh body M whose world bounds come within max_sep of the world bounds [lo,hi]; collides X with each in index
   order and groups the manifolds (sgo_mesh.h).  Returns the number of groups; normals point from the mesh to X. */
static int collide_with_mesh(const sgo_body* M, const sgo_shape* X, v3 lo, v3 hi, float max_sep, sgo_manifold* out, int active_edges, v3 movement)
{
	const m33 R = quat_to_m33(M->rot);
	sgo_mesh_contacts mc; mc.ng = 0;
	const v3 e = V3(max_sep, max_sep, max_sep);
	const v3 qlo = v3_sub(lo, e), qhi = v3_add(hi, e);
	for (uint32_t t = 0; t < M->mesh->nt; ++t) {
		const v3 a = M->mesh->verts[M->mesh->tris[3 * t]], b = M->mesh->verts[M->mesh->tris[3 * t + 1]], c = M->mesh->verts[M->mesh->tris[3 * t + 2]];
		const v3 wa = v3_add(M->pos, m33_mul(R, a)), wb = v3_add(M->pos, m33_mul(R, b)), wc = v3_add(M->pos, m33_mul(R, c));
		const v3 tmin = v3_min(v3_min(wa, wb), wc), tmax = v3_max(v3_max(wa, wb), wc);
		if (tmax.x < qlo.x || tmin.x > qhi.x || tmax.y < qlo.y || tmin.y > qhi.y || tmax.z < qlo.z || tmin.z > qhi.z) continue;
		sgo_hull th; v3 cen, n;
		sgo_tri_hull(a, b, c, &th, &cen, &n);
		sgo_hview T; T.pos = v3_add(M->pos, m33_mul(R, cen)); T.R = R; T.scale = V3(1.0f, 1.0f, 1.0f); T.h = &th;
		sgo_manifold m;
		if (sgo_collide_tri(X, &T, m33_mul(R, n), max_sep, &m, active_edges ? (unsigned)M->mesh->edges[t] : 7u, movement)) sgo_mesh_add(&mc, &m);
	}
	return sgo_mesh_finish(&mc, out);
}

/* swept sphere against mesh body M: closest front-side touch; on equal distance the lower triangle index wins */
static float cast_sphere_mesh(const sgo_body* M, v3 o, v3 d, float max_t, float rs, v3* n_out, v3* p_out)
{
	const m33 R = quat_to_m33(M->rot);
	const v3 ol = m33_tmul(R, v3_sub(o, M->pos)), dl = m33_tmul(R, d);
	float best = max_t; int hit = 0; v3 bn = V3(0, 0, 0);
	for (uint32_t t = 0; t < M->mesh->nt; ++t) {
		const v3 a = M->mesh->verts[M->mesh->tris[3 * t]], b = M->mesh->verts[M->mesh->tris[3 * t + 1]], c = M->mesh->verts[M->mesh->tris[3 * t + 2]];
		v3 nn;
		const float tt = sgo_cast_sphere_tri(ol, dl, a, b, c, best, rs, &nn);
		if (tt >= 0.0f && (tt < best || !hit)) { best = tt; hit = 1; bn = nn; }
	}
	if (!hit) return -1.0f;
	const v3 n = m33_mul(R, bn);
	*n_out = n;
	*p_out = v3_sub(v3_add(o, v3_scale(d, best)), v3_scale(n, rs));
	return best;
}

/* the wheel itself (sgo_cast_disc) against mesh body M: the search per triangle whose bounds the swept wheel can reach; closest touch, on equal distance the lower
   triangle index wins */
typedef struct { v3 d; v3 a, b, c; float max_t, rho; } disc_tri_ctx;
static float disc_probe_tri(const void* ctx, v3 start, v3* n_out, v3* p_out)
{
	const disc_tri_ctx* c = (const disc_tri_ctx*)ctx;
	const float t = sgo_cast_sphere_tri(start, c->d, c->a, c->b, c->c, c->max_t, c->rho, n_out);
	if (t >= 0.0f) *p_out = v3_sub(v3_add(start, v3_scale(c->d, t)), v3_scale(*n_out, c->rho));
	return t;
}
static float cast_disc_mesh(const sgo_body* M, v3 o, v3 d, v3 e, v3 din, float disc_r, float rho, float max_t, v3* n_out, v3* p_out)
{
	const m33 R = quat_to_m33(M->rot);
	const v3 ol = m33_tmul(R, v3_sub(o, M->pos)), dl = m33_tmul(R, d), el = m33_tmul(R, e), dinl = m33_tmul(R, din);
	const v3 end = v3_add(ol, v3_scale(dl, max_t));
	const float m = disc_r + rho + 2.0e-3f;
	const v3 lo = v3_sub(v3_min(ol, end), V3(m, m, m)), hi = v3_add(v3_max(ol, end), V3(m, m, m));
	float best = 0.0f; int hit = 0; v3 bn = V3(0, 0, 0), bp = V3(0, 0, 0);
	disc_tri_ctx c; c.d = dl; c.max_t = max_t; c.rho = rho;
	for (uint32_t t = 0; t < M->mesh->nt; ++t) {
		c.a = M->mesh->verts[M->mesh->tris[3 * t]]; c.b = M->mesh->verts[M->mesh->tris[3 * t + 1]]; c.c = M->mesh->verts[M->mesh->tris[3 * t + 2]];
		const v3 tlo = v3_min(v3_min(c.a, c.b), c.c), thi = v3_max(v3_max(c.a, c.b), c.c);
		if (thi.x < lo.x || tlo.x > hi.x || thi.y < lo.y || tlo.y > hi.y || thi.z < lo.z || tlo.z > hi.z) continue;
		v3 nn, pp;
		const float tt = sgo_cast_disc(disc_probe_tri, &c, ol, el, dinl, disc_r, &nn, &pp);
		if (tt >= 0.0f && (!hit || tt < best)) { best = tt; hit = 1; bn = nn; bp = pp; }
	}
	if (!hit) return -1.0f;
	*n_out = m33_mul(R, bn);
	*p_out = v3_add(M->pos, m33_mul(R, bp));
	return best;
}

/* ConvexHullShapeSettings::Create */
SGO_API int sgo_hull_create_com(sgo_world* w, const float* pts, uint32_t n, const float* com_offset, sgp_hull_info* info)
{
	if (!w || !pts || !info || n < 4 || n > 100000) return SGP_ERR_INVALID;
	sgo_hull* h = (sgo_hull*)malloc(sizeof(sgo_hull));
	float com[3], rot[4];
	if (sgo_hull_build(pts, (int)n, com_offset, h, com, rot) != 0) { free(h); return SGP_ERR_REJECTED; }
	uint32_t id;
	if (w->n_free_hull_ids) id = w->free_hull_ids[--w->n_free_hull_ids];
	else {
		if (w->n_hulls == w->cap_hulls) { w->cap_hulls *= 2; w->hulls = (sgo_hull**)realloc(w->hulls, sizeof(sgo_hull*) * w->cap_hulls); }
		id = w->n_hulls++;
	}
	w->hulls[id] = h;
	memset(info, 0, sizeof(*info));
	info->hull_id = id; info->num_vertices = (uint32_t)h->nv; info->num_faces = (uint32_t)h->nf; info->num_edges = (uint32_t)h->ne;
	memcpy(info->com, com, sizeof(com)); memcpy(info->rot, rot, sizeof(rot));
	info->volume = h->volume;
	info->unit_inertia[0] = h->unit_inertia.x; info->unit_inertia[1] = h->unit_inertia.y; info->unit_inertia[2] = h->unit_inertia.z;
	info->aabb_min[0] = h->aabb_min.x; info->aabb_min[1] = h->aabb_min.y; info->aabb_min[2] = h->aabb_min.z;
	info->aabb_max[0] = h->aabb_max.x; info->aabb_max[1] = h->aabb_max.y; info->aabb_max[2] = h->aabb_max.z;
	return SGP_OK;
}

SGO_API int sgo_hull_create(sgo_world* w, const float* pts, uint32_t n, sgp_hull_info* info) { return sgo_hull_create_com(w, pts, n, NULL, info); }

/* Test hook: the hull as stored (body frame). verts[nv][3], planes[nf][4]; returns nv | nf << 16. */
SGO_API int sgo_hull_dump(sgo_world* w, uint32_t id, float* verts, float* planes)
{
	if (!w || id >= w->n_hulls) return -1;
	const sgo_hull* h = w->hulls[id];
	for (int i = 0; i < h->nv; ++i) { verts[3 * i] = h->verts[i].x; verts[3 * i + 1] = h->verts[i].y; verts[3 * i + 2] = h->verts[i].z; }
	for (int f = 0; f < h->nf; ++f) { planes[4 * f] = h->normals[f].x; planes[4 * f + 1] = h->normals[f].y; planes[4 * f + 2] = h->normals[f].z; planes[4 * f + 3] = h->plane_d[f]; }
	return h->nv | (h->nf << 16);
}

/* The wheel cast against one body desc (test hook for sgo_cast_disc_body): disc of radius disc_r in the plane spanned by e and din, rounded by rho. */
SGO_API float sgo_cast_disc_hook(const sgp_body_desc* b, const float o[3], const float d[3], const float e[3], const float din[3], float disc_r, float rho, float max_t, float* n_out, float* p_out)
{
	const quat q = { b->rot[0], b->rot[1], b->rot[2], b->rot[3] };
	v3 n = V3(0, 0, 0), p = V3(0, 0, 0);
	const float t = sgo_cast_disc_body(b->shape_type, b->shape, NULL, V3(b->pos[0], b->pos[1], b->pos[2]), quat_to_m33(q), V3(o[0], o[1], o[2]), V3(d[0], d[1], d[2]), V3(e[0], e[1], e[2]), V3(din[0], din[1], din[2]), disc_r, rho, max_t, &n, &p);
	if (n_out) { n_out[0] = n.x; n_out[1] = n.y; n_out[2] = n.z; }
	if (p_out) { p_out[0] = p.x; p_out[1] = p.y; p_out[2] = p.z; }
	return t;
}

/* Sphere cast against one body desc (test hook for sgo_cast_sphere_body). Returns t or -1. */
SGO_API float sgo_cast_sphere(const sgp_body_desc* b, const float o[3], const float d[3], float max_t, float rs, float* n_out, float* p_out)
{
	const quat q = { b->rot[0], b->rot[1], b->rot[2], b->rot[3] };
	v3 n = V3(0, 0, 0), p = V3(0, 0, 0);
	const float t = sgo_cast_sphere_body(b->shape_type, b->shape, NULL, V3(b->pos[0], b->pos[1], b->pos[2]), quat_to_m33(q), V3(o[0], o[1], o[2]), V3(d[0], d[1], d[2]), max_t, rs, &n, &p);
	n_out[0] = n.x; n_out[1] = n.y; n_out[2] = n.z; p_out[0] = p.x; p_out[1] = p.y; p_out[2] = p.z;
	return t;
}

/* Constraints of the step just taken (the contact cache): key a<<32|b, colour, np, lambdas, normal. */
typedef struct { uint32_t a, b; int32_t colour; int32_t np; float n[3]; float lam_n[4]; float lam_t1[4]; float lam_t2[4]; float bias[4]; } sgo_constraint_dump;
SGO_API int sgo_world_dump_constraints(sgo_world* w, sgo_constraint_dump* out, uint32_t cap, uint32_t* n_out)
{
	uint32_t n = 0;
	for (uint32_t k = 0; k < w->n_prev; ++k) {
		const sgo_constraint* c = &w->prev[w->prev_idx_sorted[k]];
		if (c->carried) continue;      /* (kept for a sleeping pair: not a constraint of the step) */
		if (n >= cap) { ++n; continue; }
		sgo_constraint_dump* d = &out[n++];
		memset(d, 0, sizeof(*d));
		d->a = c->a; d->b = c->b; d->colour = c->colour; d->np = c->np;
		d->n[0] = c->n.x; d->n[1] = c->n.y; d->n[2] = c->n.z;
		for (int i = 0; i < c->np; ++i) { d->lam_n[i] = c->pt[i].lam_n; d->lam_t1[i] = c->pt[i].lam_t1; d->lam_t2[i] = c->pt[i].lam_t2; d->bias[i] = c->pt[i].bias; }
	}
	*n_out = n;
	return SGP_OK;
}

/* CharacterVirtual's CollideShape: every body within max_separation of a capsule (brute force over the bodies whose bounds
   overlap the capsule's). */
static int cmp_query_contact(const void* a, const void* b)
{
	const sgp_query_contact* x = (const sgp_query_contact*)a; const sgp_query_contact* y = (const sgp_query_contact*)b;
	if (x->query != y->query) return x->query < y->query ? -1 : 1;
	if (x->body != y->body) return x->body < y->body ? -1 : 1;
	return x->sub_shape < y->sub_shape ? -1 : (x->sub_shape > y->sub_shape ? 1 : 0);      /* (the point index, until the final pass below) */
}
SGO_API int sgo_collide_capsules(sgo_world* w, const sgp_capsule_query* qs, uint32_t n, sgp_query_contact* out, uint32_t cap, uint32_t* n_out)
{
	uint32_t cnt = 0;
	for (uint32_t k = 0; k < n; ++k) {
		const sgp_capsule_query* q = &qs[k];
		sgo_shape sc; memset(&sc, 0, sizeof(sc));
		sc.pos = V3(q->pos[0], q->pos[1], q->pos[2]);
		const quat qq = { q->rot[0], q->rot[1], q->rot[2], q->rot[3] };
		sc.R = quat_to_m33(qq); sc.type = SGO_SHAPE_CAPSULE; sc.p[0] = q->radius; sc.p[1] = q->half_height; sc.hull = NULL;
		const v3 ax = v3_scale(sc.R.c2, q->half_height);
		const float e = q->radius + q->max_separation;
		const v3 ext = V3(fabsf(ax.x) + e, fabsf(ax.y) + e, fabsf(ax.z) + e);
		const v3 lo = v3_sub(sc.pos, ext), hi = v3_add(sc.pos, ext);
		for (uint32_t j = 0; j < w->high; ++j) {
			const sgo_body* b = &w->bodies[j];
			if (!b->alive || b->is_alias || j == q->ignore_id) continue;
			if (q->collidable_only && !(b->layer == SGP_LAYER_NON_MOVING || b->layer == SGP_LAYER_MOVING)) continue;
			if (b->aabb_max.x < lo.x || b->aabb_min.x > hi.x || b->aabb_max.y < lo.y || b->aabb_min.y > hi.y || b->aabb_max.z < lo.z || b->aabb_min.z > hi.z) continue;
			const sgo_shape sb = body_shape_xf(b);
			sgo_manifold mm[SGO_MESH_MAX_GROUPS]; int ng;
			if (b->shape_type == SGP_SHAPE_MESH) {
				/* CharacterVirtual::GetContactsAtPosition: mActiveEdgeMode = CollideOnlyWithActive, mActiveEdgeMovementDirection = the direction of travel */
				ng = collide_with_mesh(b, &sc, lo, hi, q->max_separation, mm, q->active_edges != 0 && g_active_edges, V3(q->movement[0], q->movement[1], q->movement[2]));
			}
			else ng = sgo_collide(&sb, &sc, q->max_separation, &mm[0]) ? 1 : 0;        /* normal from the body to the capsule */
			for (int g = 0; g < ng; ++g) { const sgo_manifold m = mm[g];
			for (int i = 0; i < m.np; ++i) {
				if (cnt < cap) {
					sgp_query_contact* c = &out[cnt];
					memset(c, 0, sizeof(*c));
					c->query = k; c->body = j; c->sub_shape = (uint32_t)(4 * g + i);
					c->point[0] = m.p1[i].x; c->point[1] = m.p1[i].y; c->point[2] = m.p1[i].z;
					c->normal[0] = m.n.x; c->normal[1] = m.n.y; c->normal[2] = m.n.z;
					c->distance = v3_dot(v3_sub(m.p2[i], m.p1[i]), m.n);
					const v3 pv = b->motion == SGP_MOTION_STATIC ? V3(0, 0, 0) : v3_add(b->linv, v3_cross(b->angv, v3_sub(m.p1[i], b->pos)));
					c->point_velocity[0] = pv.x; c->point_velocity[1] = pv.y; c->point_velocity[2] = pv.z;
					c->motion_type = (uint32_t)b->motion; c->is_sensor = (uint32_t)b->is_sensor; c->inv_mass = b->inv_mass; c->userdata = b->userdata;
				}
				++cnt;
			}
			}
		}
	}
	*n_out = cnt;
	qsort(out, cnt < cap ? cnt : cap, sizeof(sgp_query_contact), cmp_query_contact);
	for (uint32_t i = 0; i < (cnt < cap ? cnt : cap); ++i) out[i].body = compound_id_of(w, out[i].body, &out[i].sub_shape);       /* a compound's children report as the compound */
	return SGP_OK;
}

SGO_API int sgo_spherecast(sgo_world* w, const sgp_ray* rays, const float* radii, uint32_t n, sgp_hit* hits)
{
	for (uint32_t k = 0; k < n; ++k) {
		const v3 o = V3(rays[k].origin[0], rays[k].origin[1], rays[k].origin[2]);
		const v3 d = V3(rays[k].dir[0], rays[k].dir[1], rays[k].dir[2]);
		float best = rays[k].max_t; uint32_t bid = SGP_INVALID_ID; v3 bn = V3(0, 0, 0);
		for (uint32_t i = 0; i < w->high; ++i) {
			const sgo_body* b = &w->bodies[i];
			if (!b->alive || b->is_alias || i == rays[k].ignore_id || b->is_sensor) continue;
			if (rays[k].collidable_only && !(b->layer == SGP_LAYER_NON_MOVING || b->layer == SGP_LAYER_MOVING)) continue;
			if (!cast_reaches_bounds(b, o, d, radii[k], rays[k].max_t)) continue;
			v3 nn, pp;
			const float t = b->shape_type == SGP_SHAPE_MESH ? cast_sphere_mesh(b, o, d, best, radii[k], &nn, &pp)
			                                                : sgo_cast_sphere_body(b->shape_type, b->shape, b->hull, b->pos, quat_to_m33(b->rot), o, d, best, radii[k], &nn, &pp);
			if (t >= 0.0f && (t < best || bid == SGP_INVALID_ID) && t <= best) { best = t; bid = i; bn = nn; }
		}
		memset(&hits[k], 0, sizeof(hits[k]));
		hits[k].id = bid; hits[k].t = bid == SGP_INVALID_ID ? 0.0f : best;
		hits[k].normal[0] = bn.x; hits[k].normal[1] = bn.y; hits[k].normal[2] = bn.z;
		hits[k].triangle = SGP_INVALID_ID;
		hits[k].userdata = bid == SGP_INVALID_ID ? 0 : w->bodies[bid].userdata;
		if (bid != SGP_INVALID_ID) hits[k].id = compound_id_of(w, bid, &hits[k].sub_shape);
	}
	return SGP_OK;
}

/* Ray vs one body (traceRay, PhysicsWorld.cpp:1668-1725).  Returns t or -1. */
typedef struct { uint32_t tri, mat; float u, v; } ray_sub;      /* which triangle of a mesh a ray hit, its user data, barycentrics */
static float ray_body(const sgo_body* b, v3 o, v3 d, float max_t, v3* n_out, ray_sub* sub)
{
	const m33 R = quat_to_m33(b->rot);
	const v3 ol = m33_tmul(R, v3_sub(o, b->pos)), dl = m33_tmul(R, d);
	sub->tri = SGP_INVALID_ID; sub->mat = 0; sub->u = 0.0f; sub->v = 0.0f;
	if (b->shape_type == SGP_SHAPE_MESH) {
		/* closest front-facing triangle; on equal distance the lower triangle index wins */
		float best = max_t; int hit = 0; v3 bn = V3(0, 0, 0);
		for (uint32_t t = 0; t < b->mesh->nt; ++t) {
			const v3 pa = b->mesh->verts[b->mesh->tris[3 * t]], pb = b->mesh->verts[b->mesh->tris[3 * t + 1]], pc = b->mesh->verts[b->mesh->tris[3 * t + 2]];
			float uv[2];
			const float tt = sgo_ray_tri_uv(ol, dl, pa, pb, pc, best, uv);
			if (tt >= 0.0f && (tt < best || !hit)) {
				best = tt; hit = 1; const v3 nn = v3_cross(v3_sub(pb, pa), v3_sub(pc, pa)); bn = v3_scale(nn, 1.0f / v3_len(nn));
				sub->tri = t; sub->mat = b->mesh->mats[t]; sub->u = uv[0]; sub->v = uv[1];
			}
		}
		if (!hit) return -1.0f;
		*n_out = m33_mul(R, bn);
		return best;
	}
	if (b->shape_type == SGP_SHAPE_HULL) {
		v3 nl;
		const float t = sgo_ray_hull(b->hull, ol, dl, max_t, 0.0f, &nl);
		if (t < 0.0f) return -1.0f;
		*n_out = m33_mul(R, nl);
		return t;
	}
	if (b->shape_type == SGP_SHAPE_SPHERE) {
		const float r = b->shape[0];
		const float B = v3_dot(ol, dl), C = v3_len_sq(ol) - r * r;
		if (C <= 0.0f) { *n_out = v3_neg(d); return 0.0f; }
		const float disc = B * B - C;
		if (disc < 0.0f) return -1.0f;
		const float t = -B - sqrtf(disc);
		if (t < 0.0f || t > max_t) return -1.0f;
		*n_out = m33_mul(R, v3_scale(v3_add(ol, v3_scale(dl, t)), 1.0f / r));
		return t;
	}
	if (b->shape_type == SGP_SHAPE_BOX) {
		const v3 h = V3(b->shape[0], b->shape[1], b->shape[2]);
		float t0 = 0.0f, t1 = max_t; int ax = -1; float sg = 0.0f;
		for (int k = 0; k < 3; ++k) {
			const float ok = v3_get(ol, k), dk = v3_get(dl, k), hk = v3_get(h, k);
			if (fabsf(dk) < 1.0e-12f) { if (ok < -hk || ok > hk) return -1.0f; continue; }
			float ta = (-hk - ok) / dk, tb = (hk - ok) / dk; float s = -1.0f;
			if (ta > tb) { const float tmp = ta; ta = tb; tb = tmp; s = 1.0f; }
			if (ta > t0) { t0 = ta; ax = k; sg = s; }
			if (tb < t1) t1 = tb;
			if (t0 > t1) return -1.0f;
		}
		if (ax < 0) { *n_out = v3_neg(d); return 0.0f; }
		v3 nl = V3(0, 0, 0); v3_set(&nl, ax, sg);
		*n_out = m33_mul(R, nl);
		return t0;
	}
	/* capsule along local z: infinite cylinder clipped to |z| <= hh, plus the two end spheres */
	{
		const float r = b->shape[0], hh = b->shape[1];
		{	/* starting inside comes first (else an interior cap-sphere entry can win, depending on max_t; see sgo_ray_capsule_z) */
			const v3 q = sgo_closest_on_segment(V3(0, 0, -hh), V3(0, 0, hh), ol);
			if (v3_len_sq(v3_sub(ol, q)) <= r * r) { *n_out = v3_neg(d); return 0.0f; }
		}
		float best = -1.0f; v3 bn = V3(0, 0, 0);
		const float a = dl.x * dl.x + dl.y * dl.y;
		const float bq = ol.x * dl.x + ol.y * dl.y, c = ol.x * ol.x + ol.y * ol.y - r * r;
		if (a > 1.0e-12f) {
			const float disc = bq * bq - a * c;
			if (disc >= 0.0f) {
				const float t = (-bq - sqrtf(disc)) / a;
				const float z = ol.z + dl.z * t;
				if (t >= 0.0f && t <= max_t && fabsf(z) <= hh) { best = t; bn = V3((ol.x + dl.x * t) / r, (ol.y + dl.y * t) / r, 0.0f); }
			}
		}
		for (int sgn = -1; sgn <= 1; sgn += 2) {
			const v3 oc = V3(ol.x, ol.y, ol.z - (float)sgn * hh);
			const float B = v3_dot(oc, dl), C = v3_len_sq(oc) - r * r;
			const float disc = B * B - C;
			if (disc < 0.0f) continue;
			const float t = -B - sqrtf(disc);
			if (t < 0.0f || t > max_t) continue;
			if (best < 0.0f || t < best) { best = t; bn = v3_scale(v3_add(oc, v3_scale(dl, t)), 1.0f / r); }
		}
		if (best < 0.0f) return -1.0f;
		*n_out = m33_mul(R, bn);
		return best;
	}
}

SGO_API int sgo_raycast(sgo_world* w, const sgp_ray* rays, uint32_t n, sgp_hit* hits)
{
	for (uint32_t k = 0; k < n; ++k) {
		const v3 o = V3(rays[k].origin[0], rays[k].origin[1], rays[k].origin[2]);
		const v3 d = V3(rays[k].dir[0], rays[k].dir[1], rays[k].dir[2]);
		float best = rays[k].max_t; uint32_t bid = SGP_INVALID_ID; v3 bn = V3(0, 0, 0);
		ray_sub bsub; bsub.tri = SGP_INVALID_ID; bsub.mat = 0; bsub.u = bsub.v = 0.0f;
		for (uint32_t i = 0; i < w->high; ++i) {
			const sgo_body* b = &w->bodies[i];
			if (!b->alive || b->is_alias || i == rays[k].ignore_id) continue;
			if (rays[k].collidable_only && !(b->layer == SGP_LAYER_NON_MOVING || b->layer == SGP_LAYER_MOVING)) continue;
			v3 nn; ray_sub sub;
			const float t = ray_body(b, o, d, best, &nn, &sub);
			/* closest hit; on equal t the lower body id wins (ids are visited in ascending order) */
			if (t >= 0.0f && (t < best || bid == SGP_INVALID_ID) && t <= best) { best = t; bid = i; bn = nn; bsub = sub; }
		}
		memset(&hits[k], 0, sizeof(hits[k]));
		hits[k].id = bid; hits[k].t = bid == SGP_INVALID_ID ? 0.0f : best;
		hits[k].normal[0] = bn.x; hits[k].normal[1] = bn.y; hits[k].normal[2] = bn.z;
		hits[k].triangle = bsub.tri; hits[k].material = bsub.mat; hits[k].bary[0] = bsub.u; hits[k].bary[1] = bsub.v;
		hits[k].userdata = bid == SGP_INVALID_ID ? 0 : w->bodies[bid].userdata;
		if (bid != SGP_INVALID_ID) hits[k].id = compound_id_of(w, bid, &hits[k].sub_shape);
	}
	return SGP_OK;
}
