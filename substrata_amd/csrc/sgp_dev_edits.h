// sgp_dev_edits.h -- AABB refresh and activation of an edited body.
// Device-inline functions only (no kernels), shared between stage files; included through sgp_dev_all.h, whose order is the dependency order.
#pragma once

// ---------------------------------------------------------------------------------------------------------------
// host edits: one thread per body, its commands applied in submission order

SGP_DEV void refresh_aabb(const DV& d, uint32_t i, uint32_t f)
{
	v3 mn, mx;
	compute_aabb(d, f_shape(f), d.pose[POSE_F4 * (size_t)i + 3], V3(d.pose[POSE_F4 * (size_t)i]), Q4(d.pose[POSE_F4 * (size_t)i + 1]), mn, mx);
	d.aabb_min[i] = F4(mn, 0.0f); d.aabb_max[i] = F4(mx, 0.0f);
}

SGP_DEV uint32_t activate_body(const DV& d, uint32_t i, uint32_t f)
{
	if (!(f & BF_ALIVE) || f_motion(f) == SGP_MOTION_STATIC) return f;
	if (!(f & BF_ACTIVE)) { f |= BF_ACTIVE; push_event(d.ev_activated, &d.evc->n_activated, d.cap_bodies, i); }
	reset_sleep(d, i, f_shape(f), d.pose[POSE_F4 * (size_t)i + 3], V3(d.pose[POSE_F4 * (size_t)i]), Q4(d.pose[POSE_F4 * (size_t)i + 1]));
	return f;
}
