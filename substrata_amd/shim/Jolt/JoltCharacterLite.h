// JPH::CharacterVirtual look-alike: the kinematic character controller PlayerPhysics drives (gui_client/PlayerPhysics.cpp:64-90,
// 106-114,258-353,477-481,519-545), on top of the two batched world queries of the sgp C ABI (sgp_collide_capsules = CollideShape with a
// maximum separation, sgp_spherecast = the swept test).  Same class / member names and Jolt's default settings; no Jolt code.
//
// Algorithm (a restatement of Jolt's CharacterVirtual::Update / ExtendedUpdate from upstream knowledge; Jolt is not in the tree):
//   MoveShape: up to mMaxCollisionIterations times { contacts within predictive distance + padding -> one plane constraint per
//     contact (+ a vertical wall for slopes steeper than mMaxSlopeAngle) -> SolveConstraints: advance to the earliest plane, slide
//     along it (along the crease of two planes; stop in a corner of three), OnContactSolve may veto -> swept test of the displacement }
//   UpdateSupportingContact: ground state / normal / velocity / body from the touching contacts inside the supporting volume
//   ExtendedUpdate: + StickToFloor (walking off a small ledge / down a slope keeps contact) + WalkStairs (step up, forward, down)
//   dynamic bodies in the way are pushed with at most mMaxStrength (an impulse applied before the next physics step).
// The character capsule keeps its axis along mUp (identity rotation), which is how PlayerPhysics uses it.
#pragma once
#include "JoltLite.h"
#include "../../../include/sgp.h"
#include <cmath>
#include <cfloat>
#include <vector>
#include <memory>
#include <algorithm>

namespace JPH
{
	typedef Vec3 Vec3Arg; typedef Vec3 RVec3Arg; typedef Quat QuatArg;
	class PhysicsMaterial {};
	class TempAllocator {};
	class BodyFilter { public: virtual ~BodyFilter() {} virtual uint32_t ignored() const { return 0xFFFFFFFFu; } };
	class IgnoreSingleBodyFilter : public BodyFilter { public: explicit IgnoreSingleBodyFilter(const BodyID& id) : body(id) {} uint32_t ignored() const override { return body.GetIndex(); } BodyID body; };
	class ShapeFilter {};
	class Plane { public: Plane() : n(0, 0, 1), c(1.0e10f) {} Plane(const Vec3& normal, float constant) : n(normal), c(constant) {} float SignedDistance(const Vec3& p) const { return n.x * p.x + n.y * p.y + n.z * p.z + c; } Vec3 n; float c; };

	// the two shapes PlayerPhysics builds: a capsule, moved so that the character position is at its bottom (PlayerPhysics.cpp:66-76)
	class CharacterShape { public: virtual ~CharacterShape() {} float radius = 0.3f, half_height = 0.65f; Vec3 offset; };
	class CapsuleShape : public CharacterShape { public: CapsuleShape(float half_height_of_cylinder, float r) { radius = r; half_height = half_height_of_cylinder; } };
	template <class T> class CharRef
	{
	public:
		CharRef() {} CharRef(T* p) : ptr(p) {} template <class U> CharRef(const CharRef<U>& o) : ptr(o.shared()) {}
		T* operator->() const { return ptr.get(); } T* GetPtr() const { return ptr.get(); } operator T*() const { return ptr.get(); }
		CharRef& operator=(T* p) { ptr.reset(p); return *this; } bool operator==(const CharRef& o) const { return ptr == o.ptr; }
		const std::shared_ptr<T>& shared() const { return ptr; }
	private:
		std::shared_ptr<T> ptr;
	};
	class RotatedTranslatedShapeSettings
	{
	public:
		RotatedTranslatedShapeSettings(const Vec3& position, const Quat& /*rotation: y-axis capsule -> z axis*/, CharacterShape* inner) { shape = new CharacterShape(*inner); shape->offset = position; delete inner; }
		struct Result { CharacterShape* s; CharRef<CharacterShape> Get() const { return CharRef<CharacterShape>(s); } };
		Result Create() const { return Result{ shape }; }
	private:
		CharacterShape* shape;
	};

	class CharacterContactSettings { public: bool mCanPushCharacter = true, mCanReceiveImpulses = true; };
	class CharacterVirtual;
	class CharacterContactListener
	{
	public:
		virtual ~CharacterContactListener() {}
		virtual void OnAdjustBodyVelocity(const CharacterVirtual*, const Body&, Vec3&, Vec3&) {}
		virtual bool OnContactValidate(const CharacterVirtual*, const BodyID&, const SubShapeID&) { return true; }
		virtual void OnContactAdded(const CharacterVirtual*, const BodyID&, const SubShapeID&, RVec3Arg, Vec3Arg, CharacterContactSettings&) {}
		virtual void OnContactSolve(const CharacterVirtual*, const BodyID&, const SubShapeID&, RVec3Arg, Vec3Arg, Vec3Arg, const PhysicsMaterial*, Vec3Arg, Vec3&) {}
	};

	class CharacterVirtualSettings
	{
	public:
		CharRef<CharacterShape> mShape;
		Vec3 mUp = Vec3(0, 1, 0);
		Plane mSupportingVolume;
		float mMaxSlopeAngle = 50.0f * 3.14159265f / 180.0f;
		float mMass = 70.0f, mMaxStrength = 100.0f;
		float mPredictiveContactDistance = 0.1f, mCharacterPadding = 0.02f, mPenetrationRecoverySpeed = 1.0f, mCollisionTolerance = 1.0e-3f;
		uint mMaxCollisionIterations = 5, mMaxConstraintIterations = 15;
		float mMinTimeRemaining = 1.0e-4f;
	};

	// (Jolt declares the ground state in CharacterBase, which Character and CharacterVirtual derive from: PlayerPhysics.cpp:226-233 names it there)
	class CharacterBase
	{
	public:
		enum class EGroundState { OnGround, OnSteepGround, NotSupported, InAir };
	};
	class CharacterVirtual : public CharacterBase
	{
	public:
		struct ExtendedUpdateSettings
		{
			Vec3 mStickToFloorStepDown = Vec3(0, -0.5f, 0), mWalkStairsStepUp = Vec3(0, 0.4f, 0);
			float mWalkStairsMinStepForward = 0.02f, mWalkStairsStepForwardTest = 0.15f, mWalkStairsCosAngleForwardContact = 0.2588f;
			Vec3 mWalkStairsStepDownExtra = Vec3(0, 0, 0);
		};
		struct Contact { BodyID body; uint32_t sub_shape; Vec3 point, normal, velocity; float distance; bool sensor, dynamic; float inv_mass; uint64_t userdata; uint64_t key() const { return ((uint64_t)body.GetIndex() << 32) | sub_shape; } };

		CharacterVirtual(const CharacterVirtualSettings* s, RVec3Arg position, QuatArg /*rotation*/, PhysicsSystem* system)
			: settings(*s), shape(s->mShape), position(position), physics_system(system), world(system->world) { cos_max_slope = std::cos(settings.mMaxSlopeAngle); }
		CharacterVirtual(const CharRef<CharacterVirtualSettings>& s, RVec3Arg position, QuatArg rotation, PhysicsSystem* system) : CharacterVirtual(s.GetPtr(), position, rotation, system) {}

		void SetListener(CharacterContactListener* l) { listener = l; }
		RVec3 GetPosition() const { return position; }
		void SetPosition(RVec3Arg p) { position = p; }
		Vec3 GetLinearVelocity() const { return linear_velocity; }
		void SetLinearVelocity(Vec3Arg v) { linear_velocity = v; }
		const CharacterShape* GetShape() const { return shape.GetPtr(); }
		bool SetShape(const CharRef<CharacterShape>& s, float, const BroadPhaseLayerFilter&, const ObjectLayerFilter&, const BodyFilter&, const ShapeFilter&, TempAllocator&) { shape = s; return true; }
		EGroundState GetGroundState() const { return ground_state; }
		bool IsSupported() const { return ground_state == EGroundState::OnGround || ground_state == EGroundState::OnSteepGround; }
		Vec3 GetGroundNormal() const { return ground_normal; }
		Vec3 GetGroundVelocity() const { return ground_velocity; }
		RVec3 GetGroundPosition() const { return ground_position; }
		BodyID GetGroundBodyID() const { return ground_body; }
		bool IsSlopeTooSteep(Vec3Arg normal) const { return dot(normal, settings.mUp) < cos_max_slope; }
		const std::vector<Contact>& GetActiveContacts() const { return active; }

		// re-sample the velocity of what we stand on (PlayerPhysics.cpp:271)
		void UpdateGroundVelocity() { if (!ground_body.IsInvalid()) { std::vector<Contact> c; getContacts(position, 0xFFFFFFFFu, c); updateSupportingContact(c, false); } }

		void Update(float dt, Vec3Arg /*gravity*/, const BroadPhaseLayerFilter&, const ObjectLayerFilter&, const BodyFilter& body_filter, const ShapeFilter&, TempAllocator&)
		{
			const uint32_t ignore = body_filter.ignored();
			blocked_by_steep = false;
			moveShape(position, linear_velocity, dt, ignore, true);
			std::vector<Contact> c; getContacts(position, ignore, c);
			updateSupportingContact(c, true);
		}

		// The pieces of ExtendedUpdate under their own names: PlayerPhysics.cpp:357-446 calls them one by one (its own copy of the sequence).
		Vec3 GetUp() const { return settings.mUp; }
		// the part of `desired` that does not push into a slope too steep to stand on (only while such a slope is what we are on / against)
		Vec3 CancelVelocityTowardsSteepSlopes(Vec3Arg desired) const
		{
			if (ground_state == EGroundState::OnGround || ground_state == EGroundState::InAir) return desired;
			Vec3 v = desired;
			for (const Contact& c : active) {
				if (c.sensor || !touching(c) || !IsSlopeTooSteep(c.normal)) continue;
				const Vec3 h = c.normal - settings.mUp * dot(c.normal, settings.mUp);
				const float towards = dot(h, v), l2 = h.LengthSq();
				if (towards < 0.0f && l2 > 1.0e-12f) v = v - h * (towards / l2);
			}
			return v;
		}
		// supported, moving horizontally, and pushing into something too steep to walk up: a step to try
		bool CanWalkStairs(Vec3Arg velocity) const
		{
			if (!IsSupported()) return false;
			const Vec3 hv = velocity - settings.mUp * dot(velocity, settings.mUp);
			if (hv.LengthSq() < 1.0e-12f) return false;
			for (const Contact& c : active) if (!c.sensor && touching(c) && dot(c.normal, hv - c.velocity) < 0.0f && IsSlopeTooSteep(c.normal)) return true;
			return false;
		}
		bool StickToFloor(Vec3Arg step_down, const BroadPhaseLayerFilter&, const ObjectLayerFilter&, const BodyFilter& bf, const ShapeFilter&, TempAllocator&) { return stickToFloor(step_down, bf.ignored()); }
		bool WalkStairs(float dt, Vec3Arg step_up, Vec3Arg step_forward, Vec3Arg step_forward_test, Vec3Arg step_down_extra, const BroadPhaseLayerFilter&, const ObjectLayerFilter&,
		                const BodyFilter& bf, const ShapeFilter&, TempAllocator&) { return walkStairs(dt, step_up, step_forward, step_forward_test, step_down_extra, bf.ignored()); }

		// Update + stick to the floor when walking off an edge + climb a step when a steep face stopped the horizontal move (CharacterVirtual::ExtendedUpdate)
		void ExtendedUpdate(float dt, Vec3Arg gravity, const ExtendedUpdateSettings& ext, const BroadPhaseLayerFilter& bp, const ObjectLayerFilter& ol, const BodyFilter& bf, const ShapeFilter& sf, TempAllocator& ta)
		{
			const Vec3 up = settings.mUp, wanted = linear_velocity;
			linear_velocity = CancelVelocityTowardsSteepSlopes(wanted);
			const Vec3 before = position;
			bool left_the_ground = IsSupported();
			Update(dt, gravity, bp, ol, bf, sf, ta);
			if (IsSupported()) left_the_ground = false;
			if (left_the_ground && ext.mStickToFloorStepDown.LengthSq() > 0.0f && dot(position - before, up) / dt <= 1.0e-6f) StickToFloor(ext.mStickToFloorStepDown, bp, ol, bf, sf, ta);
			if (!(ext.mWalkStairsStepUp.LengthSq() > 0.0f)) return;
			Vec3 want_h = wanted * dt; want_h = want_h - up * dot(want_h, up);
			const float want_len = std::sqrt(want_h.LengthSq());
			if (!(want_len > 0.0f)) return;
			const Vec3 ahead = want_h * (1.0f / want_len);
			Vec3 got = position - before; got = got - up * dot(got, up);
			const float got_len = std::max(0.0f, dot(got, ahead));                       // only progress in the wanted direction counts
			if (!(got_len + 1.0e-4f < want_len) || !CanWalkStairs(wanted)) return;
			const Vec3 step_forward = ahead * std::max(ext.mWalkStairsMinStepForward, want_len - got_len);
			// where to look for a floor if the step's nose is what we land on: against the ground normal, unless that points too far off our way
			Vec3 test = ground_normal * -1.0f; test = test - up * dot(test, up);
			const float tl = std::sqrt(test.LengthSq());
			test = tl > 1.0e-6f ? test * (1.0f / tl) : ahead;
			if (dot(test, ahead) < ext.mWalkStairsCosAngleForwardContact) test = ahead;
			WalkStairs(dt, ext.mWalkStairsStepUp, step_forward, test * ext.mWalkStairsStepForwardTest, ext.mWalkStairsStepDownExtra, bp, ol, bf, sf, ta);
		}

	private:
		static float dot(const Vec3& a, const Vec3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
		static Vec3 cross(const Vec3& a, const Vec3& b) { return Vec3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
		Vec3 capsuleCentre(const Vec3& pos) const { return pos + shape->offset; }
		bool touching(const Contact& c) const { return c.distance <= settings.mCollisionTolerance + 0.01f; }      // (what updateSupportingContact counts as a contact)

		// CharacterVirtual::GetContactsAtPosition(position, movement direction, ...): CollideShape with mActiveEdgeMode = CollideOnlyWithActive and
		// mActiveEdgeMovementDirection = the direction the character travels in, so that the seams of a triangulated floor do not stop it
		void getContacts(const Vec3& pos, uint32_t ignore, std::vector<Contact>& out) const { getContacts(pos, ignore, out, linear_velocity); }
		void getContacts(const Vec3& pos, uint32_t ignore, std::vector<Contact>& out, const Vec3& movement) const
		{
			sgp_capsule_query q;
			const float ml = std::sqrt(movement.LengthSq());
			q.movement[0] = ml > 0.0f ? movement.x / ml : 0.0f; q.movement[1] = ml > 0.0f ? movement.y / ml : 0.0f; q.movement[2] = ml > 0.0f ? movement.z / ml : 0.0f;
			q.active_edges = 1;
			const Vec3 c = capsuleCentre(pos);
			q.pos[0] = c.x; q.pos[1] = c.y; q.pos[2] = c.z; q.rot[0] = q.rot[1] = q.rot[2] = 0; q.rot[3] = 1;
			q.radius = shape->radius; q.half_height = shape->half_height;
			q.max_separation = settings.mPredictiveContactDistance + settings.mCharacterPadding;
			q.ignore_id = ignore; q.collidable_only = 1;                                     // PlayerPhysicsObjectLayerFilter
			sgp_query_contact buf[64]; uint32_t n = 0;
			if (sgp_collide_capsules(world, &q, 1, buf, 64, &n) != SGP_OK) n = 0;
			out.clear();
			for (uint32_t i = 0; i < std::min<uint32_t>(n, 64); ++i) {
				Contact k;
				k.body = BodyID(buf[i].body); k.sub_shape = buf[i].sub_shape; k.point = Vec3(buf[i].point[0], buf[i].point[1], buf[i].point[2]); k.normal = Vec3(buf[i].normal[0], buf[i].normal[1], buf[i].normal[2]);
				k.velocity = Vec3(buf[i].point_velocity[0], buf[i].point_velocity[1], buf[i].point_velocity[2]);
				k.distance = buf[i].distance - settings.mCharacterPadding; k.sensor = buf[i].is_sensor != 0; k.dynamic = buf[i].motion_type == SGP_MOTION_DYNAMIC;
				k.inv_mass = buf[i].inv_mass; k.userdata = buf[i].userdata;
				out.push_back(k);
			}
		}

		struct Constraint { Vec3 n, velocity; float distance; const Contact* contact; bool steep_slope; };

		// Advance along `velocity` for at most `time`, sliding on the constraint planes.  Returns the displacement and the time used.
		Vec3 solveConstraints(Vec3 velocity, float time, std::vector<Constraint>& cs, float& time_simulated)
		{
			Vec3 displacement(0, 0, 0);
			float t_left = time;
			const Constraint* previous = nullptr;
			for (uint it = 0; it < settings.mMaxConstraintIterations && t_left > 0.0f; ++it) {
				float best_toi = t_left; Constraint* hit = nullptr;
				for (Constraint& c : cs) {
					const float vn = dot(velocity - c.velocity, c.n);
					if (vn >= -1.0e-6f) continue;                                   // not approaching this plane
					const float dist = c.distance + dot(displacement, c.n) - dot(c.velocity, c.n) * (time - t_left);
					const float toi = std::max(0.0f, dist) / -vn;
					if (toi < best_toi) { best_toi = toi; hit = &c; }
				}
				displacement = displacement + velocity * best_toi;
				t_left -= best_toi;
				if (!hit) break;
				if (hit->steep_slope) {
					// hitting a slope too steep to stand on: first cancel the horizontal speed towards it, so that sliding along its plane
					// cannot carry us up the slope (the vertical wall constraint may be reached only after this one)
					const Vec3 vertical_plane_normal = hit->n - settings.mUp * dot(hit->n, settings.mUp);
					const float towards = std::min(0.0f, dot(velocity - hit->velocity, vertical_plane_normal));
					velocity = velocity - vertical_plane_normal * (towards / vertical_plane_normal.LengthSq());
				}
				// slide: cancel the approach speed relative to the surface (+ recover penetration at mPenetrationRecoverySpeed)
				const Vec3 rel = velocity - hit->velocity;
				Vec3 new_velocity = velocity - hit->n * dot(rel, hit->n);
				if (hit->contact) {
					if (IsSlopeTooSteep(hit->n) && dot(hit->n, settings.mUp) > -0.1f && dot(rel, hit->n) < -1.0e-3f) blocked_by_steep = true;
					pushBody(*hit->contact, rel, time);
					if (listener) listener->OnContactSolve(this, hit->contact->body, physics_system->subShapeID(hit->contact->body, hit->contact->sub_shape), hit->contact->point, hit->n, hit->velocity, nullptr, velocity, new_velocity);
				}
				if (previous && previous != hit && dot(new_velocity - previous->velocity, previous->n) < -1.0e-6f) {
					// would re-enter the previous plane: move along the crease of the two
					Vec3 dir = cross(hit->n, previous->n);
					const float l2 = dir.LengthSq();
					if (l2 > 1.0e-8f) { dir = dir * (1.0f / std::sqrt(l2)); new_velocity = dir * dot(velocity, dir); }
					else new_velocity = Vec3(0, 0, 0);
					// a third plane in the way of the crease direction: corner, stop
					for (const Constraint& c : cs) if (&c != hit && &c != previous && c.distance + dot(displacement, c.n) < 1.0e-3f && dot(new_velocity - c.velocity, c.n) < -1.0e-6f) { new_velocity = Vec3(0, 0, 0); break; }
				}
				previous = hit;
				velocity = new_velocity;
				if (velocity.LengthSq() < 1.0e-12f) break;
			}
			time_simulated = time - std::max(0.0f, t_left);
			if (velocity.LengthSq() < 1.0e-12f) time_simulated = time;             // standing still also uses up the time
			last_solved_velocity = velocity;
			return displacement;
		}

		void pushBody(const Contact& c, const Vec3& rel_velocity, float dt)
		{
			if (!c.dynamic || c.inv_mass <= 0.0f || dt <= 0.0f) return;
			// impulse that would stop the character against the body, limited by what the character can exert in this time
			const float vn = -dot(rel_velocity, c.normal);
			if (vn <= 0.0f) return;
			const float impulse = std::min(settings.mMass * vn, settings.mMaxStrength * dt);
			const Vec3 f = c.normal * (-impulse / dt);
			sgp_body_activate(world, c.body.GetIndex());
			sgp_body_add_force_at(world, c.body.GetIndex(), &f.x, &c.point.x);
		}

		// fraction of `displacement` the capsule can travel (swept test with its two end spheres)
		float sweepFraction(const Vec3& pos, const Vec3& displacement, uint32_t ignore) const
		{
			const float len = std::sqrt(displacement.LengthSq());
			if (len < 1.0e-6f) return 1.0f;
			const Vec3 dir = displacement * (1.0f / len);
			const Vec3 c = capsuleCentre(pos);
			sgp_ray rays[2]; float radii[2]; sgp_hit hits[2];
			for (int k = 0; k < 2; ++k) {
				const Vec3 o = c + settings.mUp * ((k ? 1.0f : -1.0f) * shape->half_height);
				rays[k].origin[0] = o.x; rays[k].origin[1] = o.y; rays[k].origin[2] = o.z; rays[k].dir[0] = dir.x; rays[k].dir[1] = dir.y; rays[k].dir[2] = dir.z;
				rays[k].max_t = len + settings.mCharacterPadding; rays[k].ignore_id = ignore; rays[k].collidable_only = 1; radii[k] = shape->radius;
			}
			if (sgp_spherecast(world, rays, radii, 2, hits) != SGP_OK) return 1.0f;
			float travel = len;
			for (int k = 0; k < 2; ++k) if (hits[k].id != SGP_INVALID_ID && hits[k].t > 1.0e-5f) {      // (t = 0: already touching / overlapping -- that is the plane constraints' business)
				// only surfaces we move INTO stop the sweep (we may start touching or slightly inside what we slide along)
				if (hits[k].normal[0] * dir.x + hits[k].normal[1] * dir.y + hits[k].normal[2] * dir.z < -0.05f) travel = std::min(travel, std::max(0.0f, hits[k].t - settings.mCharacterPadding));
			}
			return travel / len;
		}

		void moveShape(Vec3& pos, const Vec3& velocity, float dt, uint32_t ignore, bool notify)
		{
			float time_remaining = dt;
			std::vector<Contact> contacts; std::vector<Constraint> cs;
			for (uint it = 0; it < settings.mMaxCollisionIterations && time_remaining >= settings.mMinTimeRemaining; ++it) {
				getContacts(pos, ignore, contacts, velocity);
				cs.clear();
				for (const Contact& c : contacts) {
					if (notify && listener && std::find(seen_bodies.begin(), seen_bodies.end(), c.key()) == seen_bodies.end()) {
						seen_bodies.push_back(c.key());
						CharacterContactSettings cset;
						listener->OnContactAdded(this, c.body, physics_system->subShapeID(c.body, c.sub_shape), c.point, c.normal, cset);
					}
					if (c.sensor) continue;
					Constraint k; k.n = c.normal; k.velocity = c.velocity; k.distance = c.distance; k.contact = &c; k.steep_slope = false;
					if (c.distance < 0.0f) k.velocity = k.velocity + c.normal * (-c.distance * settings.mPenetrationRecoverySpeed / dt);      // push out of penetration
					const float nu = dot(c.normal, settings.mUp);
					k.steep_slope = nu > 1.0e-3f && nu < cos_max_slope;
					cs.push_back(k);
					if (k.steep_slope) {
						// too steep to stand on: also a vertical wall, so that walking against it does not climb it
						Vec3 h = c.normal - settings.mUp * nu;
						const float hl = std::sqrt(h.LengthSq());
						if (hl > 1.0e-6f) { Constraint w; w.n = h * (1.0f / hl); w.velocity = w.n * dot(c.velocity, w.n); w.distance = c.distance / hl; w.contact = &c; w.steep_slope = false; cs.push_back(w); }
					}
				}
				float time_simulated = 0.0f;
				Vec3 displacement = solveConstraints(velocity, time_remaining, cs, time_simulated);
				displacement = displacement * sweepFraction(pos, displacement, ignore);
				pos = pos + displacement;
				time_remaining -= std::max(time_simulated, settings.mMinTimeRemaining);
				if (displacement.LengthSq() < 1.0e-10f) break;
			}
			if (notify) { seen_bodies.clear(); for (const Contact& c : contacts) seen_bodies.push_back(c.key()); }
		}

		void updateSupportingContact(const std::vector<Contact>& contacts, bool store)
		{
			if (store) active = contacts;
			const Contact* best = nullptr; bool best_steep = true; float best_up = -2.0f; bool touching = false;
			for (const Contact& c : contacts) {
				if (c.sensor || c.distance > settings.mCollisionTolerance + 0.01f) continue;
				touching = true;
				const float nu = dot(c.normal, settings.mUp);
				if (nu <= 0.0f) continue;
				if (settings.mSupportingVolume.SignedDistance(c.point - position) > 0.0f) continue;      // above the supporting volume (PlayerPhysics.cpp:84)
				const bool steep = nu < cos_max_slope;
				if (!best || (best_steep && !steep) || (steep == best_steep && nu > best_up)) { best = &c; best_steep = steep; best_up = nu; }
			}
			if (best) {
				ground_state = best_steep ? EGroundState::OnSteepGround : EGroundState::OnGround;
				ground_normal = best->normal; ground_velocity = best->velocity; ground_position = best->point; ground_body = best->body;
			} else {
				ground_state = touching ? EGroundState::NotSupported : EGroundState::InAir;
				ground_normal = Vec3(0, 0, 0); ground_velocity = Vec3(0, 0, 0); ground_body = BodyID();
			}
		}

		// cast the lower sphere of the capsule; returns travel distance or -1
		float castDown(const Vec3& pos, const Vec3& step, uint32_t ignore, Vec3* normal_out) const
		{
			const float len = std::sqrt(step.LengthSq());
			if (len < 1.0e-6f) return -1.0f;
			const Vec3 dir = step * (1.0f / len), o = capsuleCentre(pos) - settings.mUp * shape->half_height;
			sgp_ray ray; float radius = shape->radius; sgp_hit hit;
			ray.origin[0] = o.x; ray.origin[1] = o.y; ray.origin[2] = o.z; ray.dir[0] = dir.x; ray.dir[1] = dir.y; ray.dir[2] = dir.z;
			ray.max_t = len; ray.ignore_id = ignore; ray.collidable_only = 1;
			if (sgp_spherecast(world, &ray, &radius, 1, &hit) != SGP_OK || hit.id == SGP_INVALID_ID) return -1.0f;
			if (normal_out) *normal_out = Vec3(hit.normal[0], hit.normal[1], hit.normal[2]);
			return hit.t;
		}

		bool stickToFloor(const Vec3& step_down, uint32_t ignore)
		{
			Vec3 n;
			const float t = castDown(position, step_down, ignore, &n);
			if (t < 0.0f || IsSlopeTooSteep(n)) return false;
			const float len = std::sqrt(step_down.LengthSq());
			position = position + step_down * (std::max(0.0f, t - settings.mCharacterPadding) / len);
			std::vector<Contact> c; getContacts(position, ignore, c);
			updateSupportingContact(c, true);
			return true;
		}

		bool walkStairs(float dt, const Vec3& step_up, const Vec3& step_forward, const Vec3& step_forward_test, const Vec3& step_down_extra, uint32_t ignore)
		{
			// up as far as there is head room
			const Vec3 start = position;
			Vec3 up_pos = position + step_up * sweepFraction(position, step_up, ignore);
			const float risen = std::sqrt((up_pos - position).LengthSq());
			if (risen < 1.0e-3f) return false;
			// forward by the step the caller asks for (what the blocked move still owed, at least the minimum step)
			if (step_forward.LengthSq() < 1.0e-10f || !(dt > 0.0f)) return false;
			Vec3 fwd_pos = up_pos;
			const Vec3 saved_velocity = last_solved_velocity; const bool saved_blocked = blocked_by_steep;
			moveShape(fwd_pos, step_forward * (1.0f / dt), dt, ignore, false);
			blocked_by_steep = saved_blocked; last_solved_velocity = saved_velocity;
			const Vec3 moved = fwd_pos - up_pos;
			if (moved.LengthSq() < 1.0e-8f) return false;                               // the way is blocked up there as well
			// and down again onto something we can stand on
			const Vec3 down = settings.mUp * -(risen + 1.0e-3f) + step_down_extra;
			Vec3 n;
			const float t = castDown(fwd_pos, down, ignore, &n);
			if (t < 0.0f) return false;                                                  // nothing under the step: stay where we were
			if (IsSlopeTooSteep(n)) {
				// we came down on the rounded nose of the step: look step_forward_test further ahead for a floor we can stand on
				Vec3 n2;
				const float t2 = castDown(fwd_pos + step_forward_test, down, ignore, &n2);
				if (t2 < 0.0f || IsSlopeTooSteep(n2)) return false;
			}
			const float len = std::sqrt(down.LengthSq());
			const Vec3 new_pos = fwd_pos + down * (std::max(0.0f, t - settings.mCharacterPadding) / len);
			if (dot(new_pos - start, settings.mUp) < 1.0e-3f && (new_pos - start).LengthSq() < 1.0e-6f) return false;
			position = new_pos;
			std::vector<Contact> c; getContacts(position, ignore, c);
			updateSupportingContact(c, true);
			return true;
		}

		CharacterVirtualSettings settings;
		CharRef<CharacterShape> shape;
		Vec3 position, linear_velocity, last_solved_velocity;
		PhysicsSystem* physics_system; sgp_world* world;
		CharacterContactListener* listener = nullptr;
		float cos_max_slope = 0.64f;
		EGroundState ground_state = EGroundState::InAir;
		Vec3 ground_normal, ground_velocity, ground_position; BodyID ground_body;
		std::vector<Contact> active; std::vector<uint64_t> seen_bodies;      // (body, sub shape) pairs already reported through OnContactAdded
		bool blocked_by_steep = false;
	};
}
