/*
 * sgo_vehicle.h -- ORACLE (test infrastructure only): wheeled vehicle constraint.
 *
 * What it restates.  Substrata drives cars through JPH::VehicleConstraint + JPH::WheeledVehicleController with a
 * VehicleCollisionTesterCastSphere (/root/reference/gui_client/CarPhysics.cpp:62,94-231; defaults
 * /root/reference/gui_client/Scripting.cpp:315-346); both are JoltPhysics v5.3.0 classes that are not in
 * /root/reference, so -- like the rest of the oracle -- this file restates Jolt's published algorithm from upstream
 * knowledge (parity unpinned; pinned only by the analytic tests in tests/test_oracle_vehicle.py):
 *   OnStep:            per wheel sphere cast along the suspension, contact frame, anti-roll bars,
 *                      controller PostCollide (wheel spin + slip -> tyre friction, engine / clutch / gearbox /
 *                      differential with limited slip, brakes)
 *   SetupVelocity:     per wheel 4 axis rows: soft suspension spring, hard max-up stop, longitudinal, lateral
 *   WarmStart / SolveVelocity (per iteration, before the contact constraints) / SolvePosition (max-up stop only)
 *
 * Deliberate simplifications, shared with the device implementation (so they do not affect GPU-vs-oracle parity):
 *   - (round 4: the wheel rows are two-body constraints like Jolt's -- AxisConstraintPart between the chassis and the body under the
 *     wheel: a DYNAMIC ground body takes the reaction impulses of the suspension, upper-stop, longitudinal and lateral rows, enters their
 *     effective masses and is read live; tyre slip and the longitudinal target still use the contact point velocity sampled at cast time,
 *     as Jolt's WheeledVehicleController does.  Vehicles are solved in index order, like the constraints of one island in Jolt;)
 *   - (round 5: the anti-roll bar term is the bias of the wheel's suspension row, as upstream hands it over;)
 *   - the (m+1)x(m+1) implicit clutch system is solved in closed form (same linear system);
 *   - acos of the slip angle uses a fixed polynomial, the wheel angle wraps by subtraction (no libm on the step path);
 *   - no pitch/roll limit (CarPhysics / BikePhysics leave mMaxPitchRollAngle at its default pi = off);
 *   - the motorcycle lean spring is solved implicitly in its damping term (see sgo_vehicle_solve_velocity);
 *   - VehicleCollisionTesterCastCylinder (BikePhysics.cpp:227) is served by the sphere cast with the half wheel width as radius.
 */
#ifndef SGO_VEHICLE_H
#define SGO_VEHICLE_H

#include "sgo_math.h"
/* (sgo_hull is declared by sgo_collide.h / sgo_hull.h, included first) */

#define SGO_MAX_WHEELS 4
#define SGO_MAX_GEARS 8
#define SGO_VEH_PI 3.14159265358979323846f

/* one row J = [-axis, -(r1 x axis), axis, r2 x axis] between the chassis and the body under the wheel (Jolt AxisConstraintPart + SpringPart);
   the body-2 terms are zero unless that body is dynamic */
typedef struct {
	v3 r1xa;            /* r1 x axis */
	v3 iI_r1xa;         /* I1^-1 (r1 x axis) */
	v3 r2xa;            /* r2 x axis */
	v3 iI_r2xa;         /* I2^-1 (r2 x axis) */
	float eff;          /* effective mass (incl. spring softness) */
	float softness, bias;
	float lambda;
	int active;
} sgo_axis_part;

typedef struct {
	/* settings (JPH::WheelSettingsWV) */
	v3 position, suspension_dir, steering_axis, wheel_up, wheel_forward;
	float sus_min, sus_max, sus_preload, spring_freq, spring_damp;
	float radius, width, inertia, ang_damping, max_steer, max_brake_torque, max_handbrake_torque;
	float long_fric[3][2], lat_fric[3][2];
	/* state (JPH::Wheel / WheelWV) */
	float angular_velocity, angle, steer_angle, suspension_length;
	int has_contact; uint32_t contact_body;
	int ground_dynamic;                /* the body under the wheel is dynamic: the rows act on it too (VehicleConstraint::SetupVelocityConstraint, body 2) */
	v3 contact_pos, contact_normal, contact_long, contact_lat, contact_point_vel;
	float axle_plane_constant;
	float anti_roll_impulse, brake_impulse;
	float long_slip, lat_slip, comb_long_fric, comb_lat_fric;
	float ground_friction;
	sgo_axis_part suspension, max_up, longitudinal, lateral;
	/* cast request of this step (filled by pre_a, consumed by the world's cast loop) */
	v3 cast_origin, cast_dir; float cast_len;
	/* ... when the tester casts the wheel itself (SGP_VEHICLE_TESTER_CYLINDER): the rounded disc's frame -- cast_din = the cast direction within the wheel plane,
	   cast_e = the in-plane direction across it, disc_r = radius of the flat disc, cast_rho = its rounding (sgo_cast_disc) */
	v3 cast_e, cast_din; float disc_r, cast_rho;
} sgo_wheel;

typedef struct { int left, right; float ratio, left_right_split, limited_slip_ratio, engine_torque_ratio; } sgo_differential;
typedef struct { int left, right; float stiffness; } sgo_anti_roll_bar;

typedef struct {
	uint32_t body;
	int alive, active;                 /* active = chassis was awake when this step's pre-step ran */
	int num_wheels;
	sgo_wheel wheels[SGO_MAX_WHEELS];
	v3 up, forward;                    /* chassis frame */
	float cast_radius, cos_max_slope;
	int tester;                        /* SGP_VEHICLE_TESTER_*: cast a sphere of cast_radius (0: a ray), or the wheel itself */
	/* engine (JPH::VehicleEngineSettings) */
	float engine_max_torque, engine_min_rpm, engine_max_rpm, engine_inertia, engine_ang_damping;
	float engine_curve[3][2];
	float engine_rpm;
	/* transmission (auto) */
	int num_gears, num_reverse_gears;
	float gear_ratios[SGO_MAX_GEARS], reverse_gear_ratios[SGO_MAX_GEARS];
	float switch_time, clutch_release_time, switch_latency, shift_up_rpm, shift_down_rpm, clutch_strength;
	int current_gear; float clutch_friction, gear_switch_time_left, clutch_release_time_left, gear_switch_latency_time_left;
	/* differentials, anti-roll bars */
	int num_differentials; sgo_differential differentials[2]; float differential_limited_slip_ratio;
	int num_anti_roll_bars; sgo_anti_roll_bar anti_roll_bars[2];
	/* driver input */
	float in_forward, in_right, in_brake, in_handbrake;
	/* JPH::MotorcycleController (BikePhysics.cpp:197-205): lean spring towards the direction of the ground reaction */
	int is_motorcycle, lean_enabled, lean_steering_limit;
	float max_lean_angle, tan_max_lean, lean_spring_constant, lean_spring_damping, lean_integration_coefficient, lean_integration_decay, lean_smoothing;
	float gravity_len;
	v3 target_lean; float lean_integrated_delta, lean_applied_impulse;
} sgo_vehicle;

/* chassis state as the vehicle rows see it */
typedef struct { v3 pos; quat rot; v3 v, w; float im; v3 inv_inertia_local; sym33 I; } sgo_chassis;

static inline float sgo_curve3(const float c[3][2], float x)
{
	if (x <= c[0][0]) return c[0][1];
	if (x <= c[1][0]) return c[0][1] + (c[1][1] - c[0][1]) * ((x - c[0][0]) / (c[1][0] - c[0][0]));
	if (x <= c[2][0]) return c[1][1] + (c[2][1] - c[1][1]) * ((x - c[1][0]) / (c[2][0] - c[1][0]));
	return c[2][1];
}

/* acos on [0,1] (Abramowitz & Stegun 4.4.45, |err| < 7e-5 rad) */
static inline float sgo_acos01(float x)
{
	const float xc = clampf(x, 0.0f, 1.0f);
	float p = -0.0187293f;
	p = p * xc + 0.0742610f;
	p = p * xc - 0.2121144f;
	p = p * xc + 1.5707288f;
	return p * sqrtf(1.0f - xc);
}

/* acos on [-1,1] and asin on [0,1] from the same polynomial */
static inline float sgo_acos11(float x) { return x >= 0.0f ? sgo_acos01(x) : SGO_VEH_PI - sgo_acos01(-x); }
static inline float sgo_asin01(float x) { return 0.5f * SGO_VEH_PI - sgo_acos01(x); }
static inline float sgo_signf(float x) { return x < 0.0f ? -1.0f : 1.0f; }
/* Rotation angle of a unit quaternion with vector part of length sl >= 0 and scalar part w >= 0 (2 atan2(sl, w)) without libm, whose
   atan2f differs in the last bit between processors: the asin series where it converges fast, the acos polynomial elsewhere. */
static inline float sgo_quat_angle(float sl, float w)
{
	if (sl < 0.25f) {
		const float x2 = sl * sl;
		return 2.0f * (sl * (1.0f + x2 * (0.16666667f + x2 * (0.075f + x2 * (0.044642857f + x2 * 0.030381944f)))));
	}
	return 2.0f * sgo_acos11(clampf(w, -1.0f, 1.0f));
}
static inline v3 sgo_normalized_or(v3 v, v3 fallback)
{
	const float l2 = v3_len_sq(v);
	if (l2 <= 1.0e-24f) return fallback;
	return v3_scale(v, 1.0f / sqrtf(l2));
}

static inline v3 sgo_rotate_about(v3 axis_unit, float angle, v3 v)
{
	float s, c;
	sgp_sincos_poly(0.5f * angle, &s, &c);
	const quat q = { axis_unit.x * s, axis_unit.y * s, axis_unit.z * s, c };
	return m33_mul(quat_to_m33(q), v);
}

static inline v3 sgo_chassis_point_vel(const sgo_chassis* c, v3 p)
{
	return v3_add(c->v, v3_cross(c->w, v3_sub(p, c->pos)));
}

/* ---- rays and sphere casts against the three primitives (VehicleCollisionTesterCastSphere; also traceRay) ------------------ */

static inline float sgo_ray_sphere(v3 oc, v3 d, float r, float max_t, v3* n_out)
{
	const float B = v3_dot(oc, d), C = v3_len_sq(oc) - r * r;
	if (C <= 0.0f) { *n_out = v3_neg(d); return 0.0f; }
	const float disc = B * B - C;
	if (disc < 0.0f) return -1.0f;
	const float t = -B - sqrtf(disc);
	if (t < 0.0f || t > max_t) return -1.0f;
	*n_out = v3_scale(v3_add(oc, v3_scale(d, t)), 1.0f / r);
	return t;
}

static inline float sgo_ray_box(v3 ol, v3 dl, v3 h, float max_t, v3* n_out)
{
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
	if (ax < 0) { *n_out = v3_neg(dl); return 0.0f; }
	v3 nl = V3(0, 0, 0); v3_set(&nl, ax, sg);
	*n_out = nl;
	return t0;
}

/* capsule along z through the origin: radius r, half height hh */
static inline float sgo_ray_capsule_z(v3 ol, v3 dl, float r, float hh, float max_t, v3* n_out)
{
	/* starting inside comes first: from the inside, the ray would otherwise "enter" the far cap's sphere at a point in the capsule's
	   interior, and whether that counted depended on max_t (found by tools/fuzz_parity.py) */
	{
		const float zc = clampf(ol.z, -hh, hh);
		const v3 dq = V3(ol.x, ol.y, ol.z - zc);
		if (v3_len_sq(dq) <= r * r) { *n_out = v3_neg(dl); return 0.0f; }
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
	*n_out = bn;
	return best;
}

static inline v3 sgo_perm_to_z(v3 v, int axis) /* coordinates permuted so that `axis` becomes z */
{
	return axis == 0 ? V3(v.y, v.z, v.x) : (axis == 1 ? V3(v.z, v.x, v.y) : v);
}
static inline v3 sgo_perm_from_z(v3 v, int axis)
{
	return axis == 0 ? V3(v.z, v.x, v.y) : (axis == 1 ? V3(v.y, v.z, v.x) : v);
}

/* A sphere of radius rs whose centre moves from o along the unit direction d for at most max_t, against one body
   (shape type / parameters p, pose pos + R).  Returns the travel distance at first touch (0 if it starts overlapping) or -1;
   n_out = world normal at the touch point on the body (towards the sphere), p_out = world touch point on the body.
   Box: the Minkowski sum box (+) ball is covered exactly by 3 boxes grown along one axis each plus 12 edge capsules. */
static inline float sgo_cast_sphere_body(int type, const float* p, const sgo_hull* hull, v3 pos, m33 R, v3 o, v3 d, float max_t, float rs, v3* n_out, v3* p_out)
{
	const v3 ol = m33_tmul(R, v3_sub(o, pos)), dl = m33_tmul(R, d);
	float t = -1.0f; v3 nl = V3(0, 0, 0);
	if (type == SGP_SHAPE_HULL) {
		t = sgo_ray_hull(hull, ol, dl, max_t, rs, &nl);
	} else if (type == SGP_SHAPE_SPHERE) {
		t = sgo_ray_sphere(ol, dl, p[0] + rs, max_t, &nl);
	} else if (type == SGP_SHAPE_CAPSULE) {
		t = sgo_ray_capsule_z(ol, dl, p[0] + rs, p[1], max_t, &nl);
	} else {
		const v3 h = V3(p[0], p[1], p[2]);
		float lim = max_t;
		if (rs <= 0.0f) {
			t = sgo_ray_box(ol, dl, h, lim, &nl);
		} else {
			for (int k = 0; k < 3; ++k) {
				v3 hk = h; v3_set(&hk, k, v3_get(h, k) + rs);
				v3 nn;
				const float tk = sgo_ray_box(ol, dl, hk, lim, &nn);
				if (tk >= 0.0f && (t < 0.0f || tk < t)) { t = tk; nl = nn; lim = tk; }
			}
			for (int axis = 0; axis < 3; ++axis) {
				const int a1 = (axis + 1) % 3, a2 = (axis + 2) % 3;
				for (int s1 = -1; s1 <= 1; s1 += 2) for (int s2 = -1; s2 <= 1; s2 += 2) {
					v3 c = V3(0, 0, 0);
					v3_set(&c, a1, (float)s1 * v3_get(h, a1)); v3_set(&c, a2, (float)s2 * v3_get(h, a2));
					v3 nn;
					const float tk = sgo_ray_capsule_z(sgo_perm_to_z(v3_sub(ol, c), axis), sgo_perm_to_z(dl, axis), rs, v3_get(h, axis), lim, &nn);
					if (tk >= 0.0f && (t < 0.0f || tk < t)) { t = tk; nl = sgo_perm_from_z(nn, axis); lim = tk; }
				}
			}
		}
	}
	if (t < 0.0f) return -1.0f;
	const v3 n = m33_mul(R, nl);
	*n_out = n;
	*p_out = v3_sub(v3_add(o, v3_scale(d, t)), v3_scale(n, rs));
	return t;
}

/* ---- the wheel itself as the cast shape (VehicleCollisionTesterCastCylinder, BikePhysics.cpp:229) -----------------------------------------------------
   BikePhysics makes the tester with inConvexRadiusFraction = 1: Jolt's CylinderShape(half width, radius, convex radius = half width) is then, cast with
   mUseShrunkenShapeAndConvexRadius, a flat DISC of radius disc_r = radius - half width, in the wheel plane, rounded by rho = half width -- the union of the spheres
   of radius rho centred on the disc.  Its first touch is therefore the earliest first touch among those spheres, and the sphere casts (KAT'd against a marching
   reference) are the building block: no general convex cast.  The cast direction lies in the wheel plane (suspension and steering axis do, for a bike), so of every
   chord of the disc along the direction only the LEADING end can touch first: the minimum is over the leading half of the rim, p(u) = u e + sqrt(disc_r^2 - u^2) din,
   u in [-disc_r, disc_r], and t(p(u)) is convex in u against a convex body (time-to-enter of a convex set along a fixed direction is convex in the start point, and
   -sqrt(disc_r^2 - u^2) is convex): 17 samples bracket the minimum, 20 golden-section steps refine it to ~1e-5 disc_r.  Every evaluation is a full sphere cast over
   the whole travel (never shortened by what another body already returned: the path of the search must not depend on the order of the candidates); the best
   evaluation is the answer.  A cast direction with a component along the axle is treated through its in-plane part (the rim's leading half); nothing but fraction 1
   of the convex radius is offered.  Against a mesh the search runs per triangle (the minimum over several convex pieces is not convex).
   f(ctx, start, &n, &p) = travel of a sphere of radius rho from start along d to its first touch with the piece, or -1. */
typedef float (*sgo_disc_probe_fn)(const void* ctx, v3 start, v3* n_out, v3* p_out);
#define SGO_DISC_SAMPLES 17
#define SGO_DISC_REFINE  20
static inline v3 sgo_disc_start(v3 o, v3 e, v3 din, float disc_r, float u)
{
	const float s = sqrtf(fmaxf(disc_r * disc_r - u * u, 0.0f));
	return v3_add(v3_add(o, v3_scale(e, u)), v3_scale(din, s));
}
static inline float sgo_cast_disc(sgo_disc_probe_fn f, const void* ctx, v3 o, v3 e, v3 din, float disc_r, v3* n_out, v3* p_out)
{
	if (!(disc_r > 0.0f)) return f(ctx, o, n_out, p_out);
	float best = -1.0f; int bk = -1; v3 bn = V3(0, 0, 0), bp = V3(0, 0, 0);
	for (int k = 0; k < SGO_DISC_SAMPLES; ++k) {
		const float u = disc_r * ((float)k * (2.0f / (float)(SGO_DISC_SAMPLES - 1)) - 1.0f);
		v3 n, p;
		const float t = f(ctx, sgo_disc_start(o, e, din, disc_r, u), &n, &p);
		if (t >= 0.0f && (bk < 0 || t < best)) { best = t; bk = k; bn = n; bp = p; }
	}
	if (bk < 0) return -1.0f;
	if (best > 0.0f) {
		const int k0 = bk > 0 ? bk - 1 : 0, k1 = bk < SGO_DISC_SAMPLES - 1 ? bk + 1 : SGO_DISC_SAMPLES - 1;
		float lo = disc_r * ((float)k0 * (2.0f / (float)(SGO_DISC_SAMPLES - 1)) - 1.0f), hi = disc_r * ((float)k1 * (2.0f / (float)(SGO_DISC_SAMPLES - 1)) - 1.0f);
		const float g = 0.61803398875f, miss = 3.0e38f;
		float x1 = hi - g * (hi - lo), x2 = lo + g * (hi - lo);
		v3 n, p;
		float t = f(ctx, sgo_disc_start(o, e, din, disc_r, x1), &n, &p);
		float f1 = t >= 0.0f ? t : miss;
		if (t >= 0.0f && t < best) { best = t; bn = n; bp = p; }
		t = f(ctx, sgo_disc_start(o, e, din, disc_r, x2), &n, &p);
		float f2 = t >= 0.0f ? t : miss;
		if (t >= 0.0f && t < best) { best = t; bn = n; bp = p; }
		for (int it = 0; it < SGO_DISC_REFINE; ++it) {
			if (f1 <= f2) { hi = x2; x2 = x1; f2 = f1; x1 = hi - g * (hi - lo); t = f(ctx, sgo_disc_start(o, e, din, disc_r, x1), &n, &p); f1 = t >= 0.0f ? t : miss; }
			else { lo = x1; x1 = x2; f1 = f2; x2 = lo + g * (hi - lo); t = f(ctx, sgo_disc_start(o, e, din, disc_r, x2), &n, &p); f2 = t >= 0.0f ? t : miss; }
			if (t >= 0.0f && t < best) { best = t; bn = n; bp = p; }
		}
	}
	*n_out = bn; *p_out = bp;
	return best;
}
/* ... against one body that is not a mesh */
typedef struct { int type; const float* p; const sgo_hull* hull; v3 pos; m33 R; v3 d; float max_t, rho; } sgo_disc_body_ctx;
static inline float sgo_disc_probe_body(const void* ctx, v3 start, v3* n_out, v3* p_out)
{
	const sgo_disc_body_ctx* c = (const sgo_disc_body_ctx*)ctx;
	return sgo_cast_sphere_body(c->type, c->p, c->hull, c->pos, c->R, start, c->d, c->max_t, c->rho, n_out, p_out);
}
static inline float sgo_cast_disc_body(int type, const float* p, const sgo_hull* hull, v3 pos, m33 R, v3 o, v3 d, v3 e, v3 din, float disc_r, float rho, float max_t, v3* n_out, v3* p_out)
{
	sgo_disc_body_ctx c; c.type = type; c.p = p; c.hull = hull; c.pos = pos; c.R = R; c.d = d; c.max_t = max_t; c.rho = rho;
	return sgo_cast_disc(sgo_disc_probe_body, &c, o, e, din, disc_r, n_out, p_out);
}

/* ---- axis rows ------------------------------------------------------------------------------------------------------------------ */

static inline void sgo_part_deactivate(sgo_axis_part* p) { p->active = 0; p->lambda = 0.0f; p->eff = 0.0f; p->softness = 0.0f; p->bias = 0.0f; }

/* hard row, or soft row when stiffness > 0 (Jolt SpringPart::CalculateSpringPropertiesWithStiffnessAndDamping).  g = the body under the wheel when it
   is dynamic (NULL otherwise): AxisConstraintPart::TemplatedCalculateInverseEffectiveMass adds its share after body 1's. */
static inline void sgo_part_setup(sgo_axis_part* p, const sgo_chassis* c, v3 r1, const sgo_chassis* g, v3 r2, v3 axis, float dt, float C, float stiffness, float damping, float bias_in)
{
	p->r1xa = v3_cross(r1, axis);
	p->iI_r1xa = sym33_mul(c->I, p->r1xa);
	float inv_eff = c->im + v3_dot(p->r1xa, p->iI_r1xa);
	if (g) {
		p->r2xa = v3_cross(r2, axis);
		p->iI_r2xa = sym33_mul(g->I, p->r2xa);
		inv_eff = inv_eff + (g->im + v3_dot(p->r2xa, p->iI_r2xa));
	} else { p->r2xa = V3(0, 0, 0); p->iI_r2xa = V3(0, 0, 0); }
	if (!(inv_eff > 0.0f)) { sgo_part_deactivate(p); return; }
	if (stiffness > 0.0f) {
		p->softness = 1.0f / (dt * (damping + dt * stiffness));
		p->bias = bias_in + dt * stiffness * p->softness * C;      /* SpringPart: mBias = inBias + dt k softness C */
		p->eff = 1.0f / (inv_eff + p->softness);
	} else {
		p->softness = 0.0f; p->bias = bias_in;
		p->eff = 1.0f / inv_eff;
	}
	p->active = 1;
}

static inline void sgo_part_apply(const sgo_axis_part* p, sgo_chassis* c, sgo_chassis* g, v3 axis, float lambda)
{
	c->v = v3_sub(c->v, v3_scale(axis, lambda * c->im));
	c->w = v3_sub(c->w, v3_scale(p->iI_r1xa, lambda));
	if (g) {
		g->v = v3_add(g->v, v3_scale(axis, lambda * g->im));
		g->w = v3_add(g->w, v3_scale(p->iI_r2xa, lambda));
	}
}

/* ground_vel: the contact point velocity sampled at cast time, what a ground body that is not dynamic contributes */
static inline void sgo_part_solve(sgo_axis_part* p, sgo_chassis* c, sgo_chassis* g, v3 ground_vel, v3 axis, float lo, float hi)
{
	const float jv = g ? (v3_dot(axis, v3_sub(c->v, g->v)) + v3_dot(p->r1xa, c->w)) - v3_dot(p->r2xa, g->w)
	                   : v3_dot(axis, v3_sub(c->v, ground_vel)) + v3_dot(p->r1xa, c->w);
	const float lambda = p->eff * (jv - (p->softness * p->lambda + p->bias));
	const float nl = clampf(p->lambda + lambda, lo, hi);
	sgo_part_apply(p, c, g, axis, nl - p->lambda);
	p->lambda = nl;
}

/* ---- pre-step, part A: steering angle and the cast request of every wheel (VehicleConstraint::OnStep, first half) ------ */

static inline void sgo_vehicle_pre_a(sgo_vehicle* v, const sgo_chassis* c, float dt)
{
	const m33 R = quat_to_m33(c->rot);
	float lean_max_steer_factor = 0.0f, velocity_sq = 0.0f;
	if (v->is_motorcycle) {
		/* MotorcycleController::PreCollide: the wheels still hold the contacts and impulses of the previous step here */
		const v3 forward = m33_mul(R, v->forward);
		const v3 world_up = V3(0.0f, 0.0f, 1.0f);
		if (v->lean_enabled) {
			v3 tl = V3(0, 0, 0);
			for (int i = 0; i < v->num_wheels; ++i) {
				const sgo_wheel* w = &v->wheels[i];
				if (w->has_contact) tl = v3_add(tl, v3_add(v3_scale(w->contact_normal, w->suspension.lambda + w->max_up.lambda), v3_scale(w->contact_lat, w->lateral.lambda)));
			}
			tl = sgo_normalized_or(tl, world_up);
			v->target_lean = v3_add(v3_scale(v->target_lean, v->lean_smoothing), v3_scale(tl, 1.0f - v->lean_smoothing));
			v->target_lean = v3_sub(v->target_lean, v3_scale(forward, v3_dot(v->target_lean, forward)));       /* lean sideways only */
			v->target_lean = sgo_normalized_or(v->target_lean, world_up);
			v3 adj_up = v3_sub(world_up, v3_scale(forward, v3_dot(world_up, forward)));
			adj_up = sgo_normalized_or(adj_up, world_up);
			const float w_angle = -sgo_signf(v3_dot(v3_cross(v->target_lean, adj_up), forward)) * sgo_acos11(clampf(v3_dot(v->target_lean, adj_up), -1.0f, 1.0f));
			if (fabsf(w_angle) > v->max_lean_angle) v->target_lean = sgo_rotate_about(forward, sgo_signf(w_angle) * v->max_lean_angle, adj_up);
			const v3 up = m33_mul(R, v->up);
			const float d_angle = -sgo_signf(v3_dot(v3_cross(v->target_lean, up), forward)) * sgo_acos11(clampf(v3_dot(v->target_lean, up), -1.0f, 1.0f));
			v->lean_integrated_delta = v->lean_integrated_delta + d_angle * dt;
		} else {
			v->target_lean = world_up;
			v->lean_integrated_delta = 0.0f;
		}
		/* steering limit: SteerAngle <= asin(WheelBase tan(MaxLean) g / (v^2 cos(caster))) */
		float lo = 3.0e38f, hi = -3.0e38f;
		for (int i = 0; i < v->num_wheels; ++i) {
			const sgo_wheel* w = &v->wheels[i];
			const float val = v3_dot(v3_add(w->position, v3_scale(w->suspension_dir, w->sus_max)), v->forward);
			lo = fminf(lo, val); hi = fmaxf(hi, val);
		}
		lean_max_steer_factor = (hi - lo) * v->tan_max_lean * v->gravity_len;
		const float vel = v3_dot(c->v, forward);
		velocity_sq = vel * vel;
		v->lean_applied_impulse = 0.0f;
	}
	for (int i = 0; i < v->num_wheels; ++i) {
		sgo_wheel* w = &v->wheels[i];
		w->steer_angle = -v->in_right * w->max_steer;                       /* WheeledVehicleController::PreCollide */
		if (v->is_motorcycle && w->max_steer != 0.0f) {
			const float cos_caster = v3_dot(w->steering_axis, v->up);
			float steer = fabsf(v->in_right) * w->max_steer;
			if (v->lean_steering_limit && velocity_sq > 1.0e-6f && cos_caster > 1.0e-6f) {
				const float arg = lean_max_steer_factor / (velocity_sq * cos_caster);
				if (arg < 1.0f) steer = fminf(steer, sgo_asin01(arg));
			}
			w->steer_angle = -sgo_signf(v->in_right) * steer;
		}
		w->cast_origin = v3_add(c->pos, m33_mul(R, w->position));
		w->cast_dir = m33_mul(R, w->suspension_dir);
		w->cast_len = w->sus_max + w->radius - v->cast_radius;
		if (v->tester == SGP_VEHICLE_TESTER_CYLINDER) {
			/* VehicleCollisionTesterCastCylinder::Collide: the wheel's cylinder, oriented as the steered wheel, starts at the attachment point and travels
			   mSuspensionMaxLength along the suspension; the suspension length is the distance travelled (UNVERIFIED: upstream) */
			const v3 steering_axis = m33_mul(R, w->steering_axis);
			const v3 fwd = sgo_rotate_about(steering_axis, w->steer_angle, m33_mul(R, w->wheel_forward));
			const v3 upw = sgo_rotate_about(steering_axis, w->steer_angle, m33_mul(R, w->wheel_up));
			const v3 axle = sgo_normalized_or(v3_cross(upw, fwd), V3(1.0f, 0.0f, 0.0f));
			w->cast_din = sgo_normalized_or(v3_sub(w->cast_dir, v3_scale(axle, v3_dot(w->cast_dir, axle))), w->cast_dir);
			w->cast_e = v3_cross(axle, w->cast_din);
			w->cast_rho = fminf(0.5f * w->width, w->radius);
			w->disc_r = w->radius - w->cast_rho;
			w->cast_len = w->sus_max;
		}
		w->has_contact = 0; w->contact_body = 0xFFFFFFFFu; w->ground_dynamic = 0;
	}
}

/* The world's cast loop reports the accepted hit of wheel i (distance t along the cast). */
static inline void sgo_vehicle_set_hit(sgo_vehicle* v, int i, uint32_t body, float t, v3 n, v3 p, v3 ground_point_vel, float ground_friction)
{
	sgo_wheel* w = &v->wheels[i];
	w->has_contact = 1; w->contact_body = body; w->ground_dynamic = 0;      /* (the world's glue sets ground_dynamic) */
	w->contact_normal = n; w->contact_pos = p; w->contact_point_vel = ground_point_vel; w->ground_friction = ground_friction;
	w->suspension_length = v->tester == SGP_VEHICLE_TESTER_CYLINDER ? t : fmaxf(0.0f, t + v->cast_radius - w->radius);
}

/* ---- pre-step, part B: everything after the casts --------------------------------------------------------------------------- */

static inline float sgo_gear_ratio(const sgo_vehicle* v)
{
	if (v->current_gear < 0) return v->reverse_gear_ratios[-v->current_gear - 1];
	if (v->current_gear == 0) return 0.0f;
	return v->gear_ratios[v->current_gear - 1];
}

static inline void sgo_transmission_update(sgo_vehicle* v, float dt, float rpm, float forward_input, int can_shift_up)
{
	const int old_gear = v->current_gear;
	if (v->current_gear == 0 && forward_input > 0.0f) v->current_gear = 1;
	else if (v->current_gear == 0 && forward_input < 0.0f) v->current_gear = -1;
	else if (v->gear_switch_latency_time_left == 0.0f) {
		if (can_shift_up && rpm > v->shift_up_rpm) {
			if (v->current_gear < 0) { if (v->current_gear > -v->num_reverse_gears) v->current_gear--; }
			else { if (v->current_gear < v->num_gears) v->current_gear++; }
		} else if (rpm < v->shift_down_rpm) {
			if (v->current_gear < 0) { const int max_gear = forward_input != 0.0f ? -1 : 0; if (v->current_gear < max_gear) v->current_gear++; }
			else { const int min_gear = forward_input != 0.0f ? 1 : 0; if (v->current_gear > min_gear) v->current_gear--; }
		}
	}
	if (old_gear != v->current_gear) {
		v->gear_switch_time_left = old_gear != 0 ? v->switch_time : 0.0f;
		v->clutch_release_time_left = v->clutch_release_time;
		v->gear_switch_latency_time_left = v->switch_latency;
		v->clutch_friction = 0.0f;
	} else if (v->gear_switch_time_left > 0.0f) {
		v->gear_switch_time_left = fmaxf(0.0f, v->gear_switch_time_left - dt);
		v->clutch_friction = 0.0f;
	} else if (v->clutch_release_time_left > 0.0f) {
		v->clutch_release_time_left = fmaxf(0.0f, v->clutch_release_time_left - dt);
		v->clutch_friction = 1.0f - v->clutch_release_time_left / v->clutch_release_time;
	} else {
		v->clutch_friction = 1.0f;
		v->gear_switch_latency_time_left = fmaxf(0.0f, v->gear_switch_latency_time_left - dt);
	}
}

/* VehicleDifferentialSettings::CalculateTorqueRatio */
static inline void sgo_differential_split(const sgo_differential* d, float wl, float wr, float* fl, float* fr)
{
	*fl = 1.0f - d->left_right_split; *fr = d->left_right_split;
	if (d->limited_slip_ratio < 3.0e38f) {
		const float ol = fmaxf(1.0e-3f, fabsf(wl)), orr = fmaxf(1.0e-3f, fabsf(wr));
		const float omin = fminf(ol, orr), omax = fmaxf(ol, orr);
		const float alpha = fminf((omax / omin - 1.0f) / (d->limited_slip_ratio - 1.0f), 1.0f);
		const float oma = 1.0f - alpha;
		if (ol < orr) { *fl = *fl * oma + alpha; *fr = *fr * oma; }
		else { *fl = *fl * oma; *fr = *fr * oma + alpha; }
	}
}

/* Returns 1 when the chassis' sleep timer must be reset (wheels still spinning).  g[i] = the body under wheel i when it is dynamic, else NULL. */
static inline int sgo_vehicle_pre_b(sgo_vehicle* v, sgo_chassis* c, sgo_chassis* const* g, float dt)
{
	const m33 R = quat_to_m33(c->rot);
	const int nw = v->num_wheels;
	/* contact frames */
	for (int i = 0; i < nw; ++i) {
		sgo_wheel* w = &v->wheels[i];
		if (!w->has_contact) { w->suspension_length = w->sus_max; continue; }
		w->axle_plane_constant = v3_dot(w->contact_normal, v3_add(w->cast_origin, v3_scale(w->cast_dir, w->suspension_length)));
		const v3 steering_axis = m33_mul(R, w->steering_axis);
		const v3 forward = sgo_rotate_about(steering_axis, w->steer_angle, m33_mul(R, w->wheel_forward));
		v3 lat = v3_cross(forward, w->contact_normal);
		const float ll = v3_len(lat);
		lat = ll > 1.0e-12f ? v3_scale(lat, 1.0f / ll) : V3(0, 0, 0);
		w->contact_lat = lat;
		w->contact_long = v3_cross(w->contact_normal, lat);
	}
	/* anti-roll bars: the "impulse" from the suspension length difference enters the suspension row of the wheel as its bias
	   (VehicleConstraint::OnStep sets Wheel::mAntiRollBarImpulse, SetupVelocityConstraint hands it to
	   AxisConstraintPart::CalculateConstraintPropertiesWithStiffnessAndDamping as inBias -- UNVERIFIED: upstream) */
	for (int i = 0; i < nw; ++i) v->wheels[i].anti_roll_impulse = 0.0f;
	for (int k = 0; k < v->num_anti_roll_bars; ++k) {
		sgo_wheel* lw = &v->wheels[v->anti_roll_bars[k].left]; sgo_wheel* rw = &v->wheels[v->anti_roll_bars[k].right];
		if (lw->has_contact && rw->has_contact) {
			const float impulse = (rw->suspension_length - lw->suspension_length) * v->anti_roll_bars[k].stiffness * dt;
			lw->anti_roll_impulse = -impulse; rw->anti_roll_impulse = impulse;
		}
	}

	/* ---- WheeledVehicleController::PostCollide ---- */
	const float old_rpm = v->engine_rpm;
	/* WheelWV::Update: spin damping, rotation angle, slip -> tyre friction */
	for (int i = 0; i < nw; ++i) {
		sgo_wheel* w = &v->wheels[i];
		w->angular_velocity = w->angular_velocity * fmaxf(0.0f, 1.0f - w->ang_damping * dt);
		w->angle = w->angle + w->angular_velocity * dt;
		if (w->angle > 2.0f * SGO_VEH_PI) w->angle = w->angle - 2.0f * SGO_VEH_PI;
		else if (w->angle < -2.0f * SGO_VEH_PI) w->angle = w->angle + 2.0f * SGO_VEH_PI;
		if (w->has_contact) {
			v3 rel = v3_sub(sgo_chassis_point_vel(c, w->contact_pos), w->contact_point_vel);
			rel = v3_sub(rel, v3_scale(w->contact_normal, v3_dot(w->contact_normal, rel)));
			const float rel_long = v3_dot(rel, w->contact_long);
			const float denom = (rel_long < 0.0f ? -1.0f : 1.0f) * fmaxf(1.0e-3f, fabsf(rel_long));
			w->long_slip = fabsf((w->angular_velocity * w->radius - rel_long) / denom);
			const float long_fr = sgo_curve3(w->long_fric, w->long_slip);
			const float rel_len = v3_len(rel);
			w->lat_slip = rel_len < 1.0e-3f ? 0.0f : sgo_acos01(fabsf(rel_long) / rel_len);
			const float lat_fr = sgo_curve3(w->lat_fric, w->lat_slip * (180.0f / SGO_VEH_PI));
			w->comb_long_fric = sqrtf(long_fr * w->ground_friction);           /* default VehicleConstraint combine function */
			w->comb_lat_fric = sqrtf(lat_fr * w->ground_friction);
		} else {
			w->long_slip = 0.0f; w->lat_slip = 0.0f; w->comb_long_fric = 0.0f; w->comb_lat_fric = 0.0f;
		}
	}
	float forward_input = fabsf(v->in_forward) * v->clutch_friction;                 /* auto transmission: no throttle while switching */
	v->engine_rpm = v->engine_rpm * fmaxf(0.0f, 1.0f - v->engine_ang_damping * dt); /* VehicleEngine::ApplyDamping */
	const float engine_torque = forward_input * v->engine_max_torque * sgo_curve3(v->engine_curve, v->engine_rpm / v->engine_max_rpm);

	/* driven differentials and their share of the clutch torque (limited slip between differentials) */
	float dd_omega[2], dd_ratio[2]; int dd_idx[2]; int ndd = 0;
	float omin = 3.0e38f, omax = 0.0f;
	for (int k = 0; k < v->num_differentials; ++k) {
		const sgo_differential* d = &v->differentials[k];
		float avg = 0.0f; int cnt = 0;
		if (d->left >= 0) { avg = avg + v->wheels[d->left].angular_velocity; ++cnt; }
		if (d->right >= 0) { avg = avg + v->wheels[d->right].angular_velocity; ++cnt; }
		if (cnt > 0) {
			avg = fabsf(avg * d->ratio / (float)cnt);
			dd_omega[ndd] = avg; dd_ratio[ndd] = d->engine_torque_ratio; dd_idx[ndd] = k; ++ndd;
			omin = fminf(omin, avg); omax = fmaxf(omax, avg);
		}
	}
	if (v->differential_limited_slip_ratio < 3.0e38f && omax > omin) {
		float tf[2]; float sum = 0.0f;
		for (int k = 0; k < ndd; ++k) { tf[k] = (omax - dd_omega[k]) / (omax - omin); sum = sum + tf[k]; }
		for (int k = 0; k < ndd; ++k) tf[k] = tf[k] / sum;
		const float lo = fmaxf(1.0e-3f, omin), hi = fmaxf(1.0e-3f, omax);
		const float alpha = fminf((hi / lo - 1.0f) / (v->differential_limited_slip_ratio - 1.0f), 1.0f);
		for (int k = 0; k < ndd; ++k) dd_ratio[k] = (1.0f - alpha) * dd_ratio[k] + alpha * tf[k];
	}
	/* driven wheels: engine->wheel speed ratio and torque fraction */
	const float trans_ratio = sgo_gear_ratio(v);
	int dw[4]; float dw_ratio[4], dw_frac[4]; int ndw = 0;
	for (int k = 0; k < ndd; ++k) {
		const sgo_differential* d = &v->differentials[dd_idx[k]];
		const float ratio = trans_ratio * d->ratio;
		if (d->left >= 0 && d->right >= 0) {
			float fl, fr;
			sgo_differential_split(d, v->wheels[d->left].angular_velocity, v->wheels[d->right].angular_velocity, &fl, &fr);
			dw[ndw] = d->left; dw_ratio[ndw] = ratio; dw_frac[ndw] = dd_ratio[k] * fl; ++ndw;
			dw[ndw] = d->right; dw_ratio[ndw] = ratio; dw_frac[ndw] = dd_ratio[k] * fr; ++ndw;
		} else if (d->left >= 0) { dw[ndw] = d->left; dw_ratio[ndw] = ratio; dw_frac[ndw] = dd_ratio[k]; ++ndw; }
		else if (d->right >= 0) { dw[ndw] = d->right; dw_ratio[ndw] = ratio; dw_frac[ndw] = dd_ratio[k]; ++ndw; }
	}
	/* implicit clutch:  tc = tcs (we' - mean_j R_j ww_j'),  we' = we + dt (te - tc)/Ie,  ww_i' = ww_i + dt R_i F_i tc / Iw_i */
	const float rpm_to_w = 2.0f * SGO_VEH_PI / 60.0f, w_to_rpm = 60.0f / (2.0f * SGO_VEH_PI);
	int solved = 0;
	if (ndw > 0) {
		const float tcs = trans_ratio != 0.0f ? v->clutch_friction * v->clutch_strength : 0.0f;
		if (tcs > 0.0f) {
			const float we = v->engine_rpm * rpm_to_w;
			float s0 = 0.0f, bsum = 0.0f;
			for (int k = 0; k < ndw; ++k) {
				const sgo_wheel* w = &v->wheels[dw[k]];
				s0 = s0 + dw_ratio[k] * w->angular_velocity;
				bsum = bsum + dw_ratio[k] * dw_ratio[k] * dw_frac[k] / w->inertia;
			}
			const float inv_m = 1.0f / (float)ndw;
			s0 = s0 * inv_m;
			const float A = dt / v->engine_inertia, B = dt * bsum * inv_m;
			const float tc = tcs * (we + A * engine_torque - s0) / (1.0f + tcs * (A + B));
			v->engine_rpm = (we + A * (engine_torque - tc)) * w_to_rpm;
			for (int k = 0; k < ndw; ++k) {
				sgo_wheel* w = &v->wheels[dw[k]];
				w->angular_velocity = w->angular_velocity + dt * dw_ratio[k] * dw_frac[k] * tc / w->inertia;
			}
			solved = 1;
		}
	}
	if (!solved) v->engine_rpm = v->engine_rpm + w_to_rpm * engine_torque * dt / v->engine_inertia;   /* VehicleEngine::ApplyTorque */
	v->engine_rpm = clampf(v->engine_rpm, v->engine_min_rpm, v->engine_max_rpm);

	int slipping = 0;
	for (int k = 0; k < ndw; ++k) {
		const sgo_wheel* w = &v->wheels[dw[k]];
		if (dw_frac[k] > 0.0f && (!w->has_contact || w->long_slip > 0.1f)) slipping = 1;
	}
	sgo_transmission_update(v, dt, v->engine_rpm, v->in_forward, !slipping && v->engine_rpm >= old_rpm);

	/* brakes */
	for (int i = 0; i < nw; ++i) {
		sgo_wheel* w = &v->wheels[i];
		const float brake_torque = v->in_brake * w->max_brake_torque + v->in_handbrake * w->max_handbrake_torque;
		w->brake_impulse = 0.0f;
		if (brake_torque > 0.0f) {
			const float to_lock = fabsf(w->angular_velocity) * w->inertia / dt;
			if (brake_torque > to_lock) {
				w->angular_velocity = 0.0f;
				w->brake_impulse = (brake_torque - to_lock) * dt / w->radius;
			} else {
				w->angular_velocity = w->angular_velocity + (w->angular_velocity < 0.0f ? 1.0f : -1.0f) * brake_torque * dt / w->inertia;
			}
		}
	}

	/* ---- VehicleConstraint::SetupVelocityConstraint ---- */
	for (int i = 0; i < nw; ++i) {
		sgo_wheel* w = &v->wheels[i];
		if (!w->has_contact) {
			sgo_part_deactivate(&w->suspension); sgo_part_deactivate(&w->max_up); sgo_part_deactivate(&w->longitudinal); sgo_part_deactivate(&w->lateral);
			continue;
		}
		const v3 r1 = v3_sub(w->contact_pos, c->pos);
		const sgo_chassis* gb = w->ground_dynamic ? g[i] : NULL;
		const v3 r2 = gb ? v3_sub(w->contact_pos, gb->pos) : V3(0, 0, 0);
		const v3 neg_n = v3_neg(w->contact_normal);
		float lam;
		if (w->sus_max > w->sus_min) {
			/* spring stiffness from frequency / damping ratio and the effective mass at the average suspension point */
			const v3 fp = v3_add(w->position, v3_scale(w->suspension_dir, 0.5f * (w->sus_min + w->sus_max)));
			const v3 fxu = v3_cross(fp, v3_neg(v->up));
			const v3 il = c->inv_inertia_local;
			const float eff_mass = 1.0f / (c->im + (fxu.x * il.x * fxu.x + fxu.y * il.y * fxu.y + fxu.z * il.z * fxu.z));
			const float omega = 2.0f * SGO_VEH_PI * w->spring_freq;
			const float stiffness = eff_mass * (omega * omega);
			const float damping = 2.0f * eff_mass * w->spring_damp * omega;
			const float Cc = w->suspension_length - w->sus_max - w->sus_preload;
			lam = w->suspension.lambda;
			sgo_part_setup(&w->suspension, c, r1, gb, r2, neg_n, dt, Cc, stiffness, damping, w->anti_roll_impulse);
			if (w->suspension.active) w->suspension.lambda = lam;
		} else sgo_part_deactivate(&w->suspension);
		if (w->suspension_length < w->sus_min) {
			lam = w->max_up.lambda;
			sgo_part_setup(&w->max_up, c, r1, gb, r2, neg_n, dt, 0.0f, 0.0f, 0.0f, 0.0f);
			if (w->max_up.active) w->max_up.lambda = lam;
			w->suspension_length = w->sus_min;
		} else sgo_part_deactivate(&w->max_up);
		/* the longitudinal row (engine / brake force) is never warm started: its impulse starts from zero every step */
		sgo_part_setup(&w->longitudinal, c, r1, gb, r2, v3_neg(w->contact_long), dt, 0.0f, 0.0f, 0.0f, 0.0f);
		w->longitudinal.lambda = 0.0f;
		lam = w->lateral.lambda;
		sgo_part_setup(&w->lateral, c, r1, gb, r2, v3_neg(w->contact_lat), dt, 0.0f, 0.0f, 0.0f, 0.0f);
		if (w->lateral.active) w->lateral.lambda = lam;
	}
	int spinning = 0;
	for (int i = 0; i < nw; ++i) if (fabsf(v->wheels[i].angular_velocity) > 10.0f * SGO_VEH_PI / 180.0f) spinning = 1;
	return spinning;
}

/* VehicleConstraint::WarmStartVelocityConstraint */
static inline void sgo_vehicle_warm_start(sgo_vehicle* v, sgo_chassis* c, sgo_chassis* const* g)
{
	for (int i = 0; i < v->num_wheels; ++i) {
		sgo_wheel* w = &v->wheels[i];
		if (!w->has_contact) continue;
		sgo_chassis* gb = w->ground_dynamic ? g[i] : NULL;
		if (w->suspension.active) sgo_part_apply(&w->suspension, c, gb, v3_neg(w->contact_normal), w->suspension.lambda);
		if (w->max_up.active) sgo_part_apply(&w->max_up, c, gb, v3_neg(w->contact_normal), w->max_up.lambda);
		if (w->lateral.active) sgo_part_apply(&w->lateral, c, gb, v3_neg(w->contact_lat), w->lateral.lambda);
	}
}

/* VehicleConstraint::SolveVelocityConstraint + WheeledVehicleController::SolveLongitudinalAndLateralConstraints */
static inline void sgo_vehicle_solve_velocity(sgo_vehicle* v, sgo_chassis* c, sgo_chassis* const* g, float dt)
{
	const int nw = v->num_wheels;
	for (int i = 0; i < nw; ++i) {
		sgo_wheel* w = &v->wheels[i];
		if (!w->has_contact) continue;
		const v3 neg_n = v3_neg(w->contact_normal);
		sgo_chassis* gb = w->ground_dynamic ? g[i] : NULL;
		if (w->suspension.active) sgo_part_solve(&w->suspension, c, gb, w->contact_point_vel, neg_n, 0.0f, 3.0e38f);   /* pushes, never pulls */
		if (w->max_up.active) sgo_part_solve(&w->max_up, c, gb, w->contact_point_vel, neg_n, 0.0f, 3.0e38f);
	}
	float max_lat[SGO_MAX_WHEELS];
	for (int i = 0; i < nw; ++i) {
		sgo_wheel* w = &v->wheels[i];
		max_lat[i] = 0.0f;
		if (!w->has_contact) continue;
		const float sus_lambda = w->suspension.lambda + w->max_up.lambda;
		const float max_long = w->comb_long_fric * sus_lambda;
		max_lat[i] = w->comb_lat_fric * sus_lambda;
		if (!w->longitudinal.active) continue;
		sgo_chassis* gb = w->ground_dynamic ? g[i] : NULL;
		const v3 rel = v3_sub(sgo_chassis_point_vel(c, w->contact_pos), w->contact_point_vel);
		const float rel_long = v3_dot(rel, w->contact_long);
		if (w->brake_impulse != 0.0f) {
			const float bi = fminf(w->brake_impulse, max_long);
			float lo, hi;
			if (rel_long >= 0.0f) { lo = -bi; hi = 0.0f; } else { lo = 0.0f; hi = bi; }
			sgo_part_solve(&w->longitudinal, c, gb, w->contact_point_vel, v3_neg(w->contact_long), lo, hi);
		} else {
			/* impulse that brings the contact patch speed to the rolling speed of the wheel within this step */
			const float desired_w = rel_long / w->radius;
			const float lin_imp = (w->angular_velocity - desired_w) * w->inertia / w->radius;
			const float prev = w->longitudinal.lambda;
			const float lim = clampf(prev + lin_imp, -max_long, max_long);
			sgo_part_solve(&w->longitudinal, c, gb, w->contact_point_vel, v3_neg(w->contact_long), lim, lim);
			w->angular_velocity = w->angular_velocity - (w->longitudinal.lambda - prev) * w->radius / w->inertia;
		}
	}
	for (int i = 0; i < nw; ++i) {
		sgo_wheel* w = &v->wheels[i];
		if (!w->has_contact || !w->lateral.active) continue;
		sgo_part_solve(&w->lateral, c, w->ground_dynamic ? g[i] : NULL, w->contact_point_vel, v3_neg(w->contact_lat), -max_lat[i], max_lat[i]);
	}
	if (v->is_motorcycle && v->lean_enabled) {
		/* MotorcycleController::SolveLongitudinalAndLateralConstraints: lean spring (PID on the angle to the target lean), only
		   with every wheel loaded; the matching linear impulse keeps the contact patches from being swept sideways */
		int all_in_contact = 1;
		for (int i = 0; i < nw; ++i) if (!v->wheels[i].has_contact || !(v->wheels[i].suspension.lambda + v->wheels[i].max_up.lambda > 0.0f)) all_in_contact = 0;
		if (all_in_contact) {
			const m33 R = quat_to_m33(c->rot);
			const v3 forward = m33_mul(R, v->forward), up = m33_mul(R, v->up);
			const float d_angle = -sgo_signf(v3_dot(v3_cross(v->target_lean, up), forward)) * sgo_acos11(clampf(v3_dot(v->target_lean, up), -1.0f, 1.0f));
			const float ddt_angle = v3_dot(c->w, forward);
			/* Jolt re-evaluates  total = (K d - D w.f + Ki integral) dt  with the current angular velocity every iteration and applies
			   the difference to what it applied before: a fixed-point iteration that only converges while D dt (f.I^-1 f) < 1.  Here the
			   same fixed point is solved for directly (w.f without the lean impulse = ddt_angle - (f.I^-1 f) applied), which is its
			   limit when it converges and stays stable when it would not. */
			const v3 If = sym33_mul(c->I, forward);
			const float iff = v3_dot(forward, If);
			const float wf0 = ddt_angle - iff * v->lean_applied_impulse;
			const float total = (v->lean_spring_constant * d_angle - v->lean_spring_damping * wf0 + v->lean_integration_coefficient * v->lean_integrated_delta) * dt
			                    / (1.0f + v->lean_spring_damping * dt * iff);
			const v3 old_w = c->w;
			c->w = v3_add(c->w, v3_scale(If, total - v->lean_applied_impulse));
			v->lean_applied_impulse = total;
			const v3 dw = v3_sub(c->w, old_w);
			v3 lin_acc = V3(0, 0, 0); float total_lambda = 0.0f;
			for (int i = 0; i < nw; ++i) {
				const sgo_wheel* w = &v->wheels[i];
				const float lam = w->suspension.lambda + w->max_up.lambda;
				total_lambda = total_lambda + lam;
				lin_acc = v3_add(lin_acc, v3_scale(v3_cross(dw, v3_sub(w->contact_pos, c->pos)), lam));
			}
			c->v = v3_sub(c->v, v3_scale(lin_acc, 1.0f / total_lambda));      /* impulse -acc / (lambda im), times im */
		} else {
			v->lean_integrated_delta = v->lean_integrated_delta * fmaxf(0.0f, 1.0f - v->lean_integration_decay * dt);
		}
	}
}

/* VehicleConstraint::SolvePositionConstraint: the axle at minimum suspension length must stay on the outer side of the plane
   through the axle position at cast time.  Works on the poses (pos, rot) of the chassis and of a dynamic body under the wheel; the
   world-space inertias are recomputed from the poses. */
static inline void sgo_vehicle_solve_position(sgo_vehicle* v, sgo_chassis* c, sgo_chassis* const* g, float baumgarte)
{
	for (int i = 0; i < v->num_wheels; ++i) {
		sgo_wheel* w = &v->wheels[i];
		if (!w->has_contact) continue;
		const m33 R = quat_to_m33(c->rot);
		const v3 ws_dir = m33_mul(R, w->suspension_dir);
		const v3 ws_pos = v3_add(c->pos, m33_mul(R, w->position));
		const v3 min_pos = v3_add(ws_pos, v3_scale(ws_dir, w->sus_min));
		const float err = v3_dot(w->contact_normal, min_pos) - w->axle_plane_constant;
		if (err < 0.0f) {
			const v3 axis = v3_neg(w->contact_normal);
			const v3 r1 = v3_sub(w->contact_pos, c->pos);
			const sym33 I = world_inv_inertia(R, c->inv_inertia_local);
			const v3 r1xa = v3_cross(r1, axis);
			const v3 iI = sym33_mul(I, r1xa);
			float inv_eff = c->im + v3_dot(r1xa, iI);
			sgo_chassis* gb = w->ground_dynamic ? g[i] : NULL;
			v3 iI2 = V3(0, 0, 0);
			if (gb) {
				const v3 r2xa = v3_cross(v3_sub(w->contact_pos, gb->pos), axis);
				iI2 = sym33_mul(world_inv_inertia(quat_to_m33(gb->rot), gb->inv_inertia_local), r2xa);
				inv_eff = inv_eff + (gb->im + v3_dot(r2xa, iI2));
			}
			if (!(inv_eff > 0.0f)) continue;
			const float lambda = -(1.0f / inv_eff) * baumgarte * err;
			c->pos = v3_sub(c->pos, v3_scale(axis, lambda * c->im));
			c->rot = quat_add_rotation_step(c->rot, v3_scale(iI, -lambda));
			if (gb) {
				gb->pos = v3_add(gb->pos, v3_scale(axis, lambda * gb->im));
				gb->rot = quat_add_rotation_step(gb->rot, v3_scale(iI2, lambda));
			}
		}
	}
}

#endif
