// oracle_jolt.cpp -- the REAL reference arithmetic for the hot path: JoltPhysics v5.3.0 driven exactly as Substrata's PhysicsWorld drives it.
//
// TEST INFRASTRUCTURE, not product code.  It cannot be built in the authoring environment: JoltPhysics is un-vendored in the reference
// (scripts/get_libs.rb:29-40 fetches tag v5.3.0) and absent from /root/reference.  A maintainer with a Jolt checkout builds it with
//     SGP_JOLT_DIR=/path/to/JoltPhysics make -C oracle jolt_ref          (oracle/Makefile; output oracle/_ref/oracle_jolt)
// and tests/test_jolt_ref.py then compares this repo's CPU oracle with it (and bench.py --jolt-baseline times it as BASELINE.md's B1 row).
// Until then the test is skipped LOUDLY and parity stays "unpinned".
//
// What it replicates (reference file:line):
//   PhysicsWorld::init            gui_client/PhysicsWorld.cpp:250-273   Factory + RegisterTypes (default allocators: the hooks only count memory)
//   PhysicsWorld::PhysicsWorld    :462-532   PhysicsSystem::Init(cMaxBodies, 0, cMaxBodyPairs, cMaxContactConstraints, layer interfaces), gravity (0,0,-9.81);
//                                            the three capacities are PARAMETERS here (the reference's 65536 / 65536 / 10240 overflow at configs 2-5)
//   layer tables                  :85-189    4 object layers -> 2 broad-phase layers, the two filters
//   PhysicsWorld::addObject       :1169-1311 sphere r 0.5 / box half 0.5 under a ScaledShape unless scale == 1, friction / restitution clamped to [0,1],
//                                            mass >= 0.001 with EOverrideMassProperties::CalculateInertia, CreateAndAddBody(DontActivate); activateObject :1342
//   PhysicsWorld::think           :1356-1364 PhysicsSystem::Update(dt, cCollisionSteps = 1, temp allocator, job system)
//   (the buoyancy sweep :1367-1442 is Substrata's own code over Jolt's Body::GetSubmergedVolume / ApplyBuoyancyImpulse: restated when --water is given)
//
// Scene in, states out -- the formats tests/jolt_ref_io.py reads and writes:
//   scene file : u32 magic 'SGPJ', u32 n, then n records of sgp_body_desc (include/sgp.h); shape_type must be SPHERE, BOX or CAPSULE
//   dump file  : u32 magic 'SGPD', u32 n, u32 n_checkpoints, then per checkpoint: u32 step, n records of sgp_body_state (id = scene index)
//   stdout     : one JSON line {"steps":…, "seconds":…, "steps_per_s":…, "threads":…, "bodies":…, "contact_constraints_last":…}
//
//   oracle_jolt scene.bin dump.bin --steps 240 --dt 0.0166666667 --checkpoints 1,10,60,240 [--threads T] [--max-bodies N] [--max-pairs N]
//               [--max-contacts N] [--water z]
#include <Jolt/Jolt.h>
#include <Jolt/RegisterTypes.h>
#include <Jolt/Core/Factory.h>
#include <Jolt/Core/TempAllocator.h>
#include <Jolt/Core/JobSystemThreadPool.h>
#include <Jolt/Physics/PhysicsSettings.h>
#include <Jolt/Physics/PhysicsSystem.h>
#include <Jolt/Physics/Collision/Shape/BoxShape.h>
#include <Jolt/Physics/Collision/Shape/SphereShape.h>
#include <Jolt/Physics/Collision/Shape/CapsuleShape.h>
#include <Jolt/Physics/Collision/Shape/ScaledShape.h>
#include <Jolt/Physics/Collision/Shape/RotatedTranslatedShape.h>
#include <Jolt/Physics/Body/BodyCreationSettings.h>
#include <Jolt/Physics/Body/BodyActivationListener.h>

#include "../../include/sgp.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

JPH_SUPPRESS_WARNINGS

namespace Layers
{
	static constexpr JPH::ObjectLayer NON_MOVING = 0, MOVING = 1, NON_MOVING_NON_COLLIDABLE = 2, MOVING_NON_COLLIDABLE = 3, NUM_LAYERS = 4;      // PhysicsWorld.h:67-74
}
namespace BroadPhaseLayers
{
	static constexpr JPH::BroadPhaseLayer NON_MOVING(0), MOVING(1);
	static constexpr JPH::uint NUM_LAYERS = 2;                                                                                                     // PhysicsWorld.cpp:85-90
}

// PhysicsWorld.cpp:95-132
class BPLayerInterfaceImpl final : public JPH::BroadPhaseLayerInterface
{
public:
	JPH::uint GetNumBroadPhaseLayers() const override { return BroadPhaseLayers::NUM_LAYERS; }
	JPH::BroadPhaseLayer GetBroadPhaseLayer(JPH::ObjectLayer layer) const override
	{
		return (layer == Layers::MOVING || layer == Layers::MOVING_NON_COLLIDABLE) ? BroadPhaseLayers::MOVING : BroadPhaseLayers::NON_MOVING;
	}
#if defined(JPH_EXTERNAL_PROFILE) || defined(JPH_PROFILE_ENABLED)
	const char* GetBroadPhaseLayerName(JPH::BroadPhaseLayer layer) const override { return layer == BroadPhaseLayers::MOVING ? "MOVING" : "NON_MOVING"; }
#endif
};
// PhysicsWorld.cpp:135-157
class ObjectVsBroadPhaseFilter final : public JPH::ObjectVsBroadPhaseLayerFilter
{
public:
	bool ShouldCollide(JPH::ObjectLayer layer, JPH::BroadPhaseLayer bp) const override
	{
		if (layer == Layers::NON_MOVING) return bp == BroadPhaseLayers::MOVING;
		return layer == Layers::MOVING;
	}
};
// PhysicsWorld.cpp:160-189
class ObjectPairFilter final : public JPH::ObjectLayerPairFilter
{
public:
	bool ShouldCollide(JPH::ObjectLayer a, JPH::ObjectLayer b) const override
	{
		if (a == Layers::NON_MOVING) return b == Layers::MOVING;
		if (a == Layers::MOVING) return b != Layers::NON_MOVING_NON_COLLIDABLE && b != Layers::MOVING_NON_COLLIDABLE;
		return false;
	}
};

static float clamp01(float v) { return v < 0.f ? 0.f : (v > 1.f ? 1.f : v); }

struct Args
{
	std::string scene, dump;
	int steps = 240; double dt = 1.0 / 60.0;
	std::vector<int> checkpoints;
	int threads = 0;
	JPH::uint max_bodies = 0, max_pairs = 0, max_contacts = 0;
	bool water = false; float water_z = 0.f;
};

static bool parse(int argc, char** argv, Args& a)
{
	if (argc < 3) return false;
	a.scene = argv[1]; a.dump = argv[2];
	for (int i = 3; i < argc; ++i) {
		const std::string k = argv[i];
		auto next = [&]() -> const char* { return i + 1 < argc ? argv[++i] : ""; };
		if (k == "--steps") a.steps = atoi(next());
		else if (k == "--dt") a.dt = atof(next());
		else if (k == "--threads") a.threads = atoi(next());
		else if (k == "--max-bodies") a.max_bodies = (JPH::uint)atoll(next());
		else if (k == "--max-pairs") a.max_pairs = (JPH::uint)atoll(next());
		else if (k == "--max-contacts") a.max_contacts = (JPH::uint)atoll(next());
		else if (k == "--water") { a.water = true; a.water_z = (float)atof(next()); }
		else if (k == "--checkpoints") { std::string s = next(); size_t p = 0; while (p < s.size()) { size_t q = s.find(',', p); if (q == std::string::npos) q = s.size(); a.checkpoints.push_back(atoi(s.substr(p, q - p).c_str())); p = q + 1; } }
		else return false;
	}
	if (a.checkpoints.empty()) a.checkpoints.push_back(a.steps);
	return true;
}

int main(int argc, char** argv)
{
	Args args;
	if (!parse(argc, argv, args)) { fprintf(stderr, "usage: oracle_jolt scene.bin dump.bin [--steps N] [--dt s] [--checkpoints a,b,c] [--threads T] [--max-bodies N] [--max-pairs N] [--max-contacts N] [--water z]\n"); return 2; }

	// ---- scene
	FILE* f = fopen(args.scene.c_str(), "rb");
	if (!f) { fprintf(stderr, "cannot open %s\n", args.scene.c_str()); return 2; }
	uint32_t magic = 0, n = 0;
	if (fread(&magic, 4, 1, f) != 1 || fread(&n, 4, 1, f) != 1 || magic != 0x4A504753u /* 'SGPJ' */) { fprintf(stderr, "bad scene file\n"); return 2; }
	std::vector<sgp_body_desc> descs(n);
	if (n && fread(descs.data(), sizeof(sgp_body_desc), n, f) != n) { fprintf(stderr, "short scene file\n"); return 2; }
	fclose(f);

	// ---- PhysicsWorld::init, :250-273
	JPH::RegisterDefaultAllocator();
	JPH::Factory::sInstance = new JPH::Factory();
	JPH::RegisterTypes();

	// ---- PhysicsWorld::PhysicsWorld, :462-532 (capacities parameterised; defaults scale with the scene the way the reference's constants do not)
	const JPH::uint cMaxBodies = args.max_bodies ? args.max_bodies : std::max<JPH::uint>(65536u, n + 1024u);
	const JPH::uint cNumBodyMutexes = 0;
	const JPH::uint cMaxBodyPairs = args.max_pairs ? args.max_pairs : std::max<JPH::uint>(65536u, 16u * n);
	const JPH::uint cMaxContactConstraints = args.max_contacts ? args.max_contacts : std::max<JPH::uint>(10240u, 8u * n);
	JPH::TempAllocatorMalloc temp_allocator;          // (the reference's 22 MiB stack allocator, GUIClient.cpp:189, is too small beyond config 1)
	const int threads = args.threads > 0 ? args.threads : std::max(1, (int)std::thread::hardware_concurrency() - 1);
	JPH::JobSystemThreadPool job_system(JPH::cMaxPhysicsJobs, JPH::cMaxPhysicsBarriers, threads);
	BPLayerInterfaceImpl broad_phase_layer_interface;
	ObjectVsBroadPhaseFilter broad_phase_layer_filter;
	ObjectPairFilter object_layer_pair_filter;
	JPH::PhysicsSystem physics_system;
	physics_system.Init(cMaxBodies, cNumBodyMutexes, cMaxBodyPairs, cMaxContactConstraints, broad_phase_layer_interface, broad_phase_layer_filter, object_layer_pair_filter);
	physics_system.SetGravity(JPH::Vec3(0, 0, -9.81f));
	// (PhysicsSettings stay at Jolt's defaults: Substrata never calls SetPhysicsSettings)
	JPH::BodyInterface& body_interface = physics_system.GetBodyInterface();

	// ---- PhysicsWorld::addObject, :1169-1311 -- per desc; the desc carries the SCALED primitive (radius, half extents), so the unit shape and
	//      its ScaledShape decorator are rebuilt from it: sphere scale = r / 0.5, box scale = half / 0.5
	std::vector<JPH::BodyID> ids(n);
	for (uint32_t i = 0; i < n; ++i) {
		const sgp_body_desc& d = descs[i];
		const JPH::EMotionType mt = d.motion_type == SGP_MOTION_DYNAMIC ? JPH::EMotionType::Dynamic : (d.motion_type == SGP_MOTION_KINEMATIC ? JPH::EMotionType::Kinematic : JPH::EMotionType::Static);
		JPH::Ref<JPH::ShapeSettings> shape;
		if (d.shape_type == SGP_SHAPE_SPHERE) {
			JPH::Ref<JPH::SphereShapeSettings> s = new JPH::SphereShapeSettings(0.5f);
			const float sc = d.shape[0] / 0.5f;
			if (sc == 1.0f) shape = s; else shape = new JPH::ScaledShapeSettings(s, JPH::Vec3(sc, sc, sc));
		} else if (d.shape_type == SGP_SHAPE_BOX) {
			JPH::Ref<JPH::BoxShapeSettings> s = new JPH::BoxShapeSettings(JPH::Vec3(0.5f, 0.5f, 0.5f));
			const JPH::Vec3 sc(d.shape[0] / 0.5f, d.shape[1] / 0.5f, d.shape[2] / 0.5f);
			if (sc == JPH::Vec3(1, 1, 1)) shape = s; else shape = new JPH::ScaledShapeSettings(s, sc);
		} else if (d.shape_type == SGP_SHAPE_CAPSULE) {
			// PlayerPhysics / AvatarGraphics build their capsules directly (PlayerPhysics.cpp:74): Jolt's capsule runs along y, this repo's along z
			JPH::Ref<JPH::CapsuleShapeSettings> s = new JPH::CapsuleShapeSettings(d.shape[1], d.shape[0]);
			shape = new JPH::RotatedTranslatedShapeSettings(JPH::Vec3::sZero(), JPH::Quat::sRotation(JPH::Vec3::sAxisX(), 0.5f * JPH::JPH_PI), s);
		} else { fprintf(stderr, "body %u: shape type %d is not handled by oracle_jolt\n", i, d.shape_type); return 2; }
		JPH::BodyCreationSettings settings(shape, JPH::RVec3(d.pos[0], d.pos[1], d.pos[2]), JPH::Quat(d.rot[0], d.rot[1], d.rot[2], d.rot[3]), mt, (JPH::ObjectLayer)d.layer);
		settings.mIsSensor = d.is_sensor != 0;
		settings.mFriction = clamp01(d.friction);                                               // :1236
		settings.mRestitution = clamp01(d.restitution);                                         // :1237
		settings.mMassPropertiesOverride.mMass = std::max(0.001f, d.mass);                      // :1238
		settings.mOverrideMassProperties = JPH::EOverrideMassProperties::CalculateInertia;      // :1239
		settings.mLinearVelocity = JPH::Vec3(d.lin_vel[0], d.lin_vel[1], d.lin_vel[2]);
		settings.mAngularVelocity = JPH::Vec3(d.ang_vel[0], d.ang_vel[1], d.ang_vel[2]);
		settings.mGravityFactor = d.gravity_factor; settings.mLinearDamping = d.linear_damping; settings.mAngularDamping = d.angular_damping;
		settings.mAllowSleeping = d.allow_sleeping != 0;
		settings.mUserData = d.userdata;
		ids[i] = body_interface.CreateAndAddBody(settings, JPH::EActivation::DontActivate);     // :1243
		if (ids[i].IsInvalid()) { fprintf(stderr, "body %u: CreateAndAddBody failed (raise --max-bodies)\n", i); return 2; }
		if (d.activate && mt != JPH::EMotionType::Static) body_interface.ActivateBody(ids[i]);  // activateObject, :1342-1353
	}
	physics_system.OptimizeBroadPhase();

	// ---- think() x steps, :1356-1364 (+ the buoyancy sweep :1367-1442)
	FILE* out = fopen(args.dump.c_str(), "wb");
	if (!out) { fprintf(stderr, "cannot write %s\n", args.dump.c_str()); return 2; }
	const uint32_t dmagic = 0x44504753u /* 'SGPD' */, ncp = (uint32_t)args.checkpoints.size();
	fwrite(&dmagic, 4, 1, out); fwrite(&n, 4, 1, out); fwrite(&ncp, 4, 1, out);
	std::vector<sgp_body_state> states(n);
	const int cCollisionSteps = 1;                                                              // :1359
	double seconds = 0.0;
	for (int step = 1; step <= args.steps; ++step) {
		const auto t0 = std::chrono::steady_clock::now();
		physics_system.Update((float)args.dt, cCollisionSteps, &temp_allocator, &job_system);   // :1363
		if (args.water) {
			// :1367-1442 with the constants of :1384-1410
			const float fluid_density = 1020.f;
			JPH::BodyIDVector active;
			physics_system.GetActiveBodies(JPH::EBodyType::RigidBody, active);
			const JPH::BodyLockInterface& lock_iface = physics_system.GetBodyLockInterface();
			for (const JPH::BodyID& id : active) {
				JPH::BodyLockWrite lock(lock_iface, id);
				if (!lock.Succeeded()) continue;
				JPH::Body& body = lock.GetBody();
				if (body.GetMotionType() != JPH::EMotionType::Dynamic) continue;
				if (body.GetWorldSpaceBounds().mMin.GetZ() >= args.water_z) continue;
				const float volume = body.GetShape()->GetVolume();
				const float mass = 1.0f / body.GetMotionProperties()->GetInverseMass();
				const float buoyancy = fluid_density * volume / mass;                          // :1387
				const bool zero_drag = false;                                                 // (use_zero_linear_drag is a per-object flag of boats only)
				body.ApplyBuoyancyImpulse(JPH::RVec3(0, 0, args.water_z), JPH::Vec3(0, 0, 1), buoyancy, zero_drag ? 0.0f : 0.1f, 3.0f, JPH::Vec3::sZero(), JPH::Vec3(0, 0, -9.81f), (float)args.dt);
			}
		}
		seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
		if (std::find(args.checkpoints.begin(), args.checkpoints.end(), step) != args.checkpoints.end()) {
			for (uint32_t i = 0; i < n; ++i) {
				sgp_body_state& s = states[i]; memset(&s, 0, sizeof(s));
				JPH::RVec3 p; JPH::Quat q; body_interface.GetPositionAndRotation(ids[i], p, q);
				JPH::Vec3 lv, av; body_interface.GetLinearAndAngularVelocity(ids[i], lv, av);
				s.pos[0] = (float)p.GetX(); s.pos[1] = (float)p.GetY(); s.pos[2] = (float)p.GetZ();
				s.rot[0] = q.GetX(); s.rot[1] = q.GetY(); s.rot[2] = q.GetZ(); s.rot[3] = q.GetW();
				s.lin_vel[0] = lv.GetX(); s.lin_vel[1] = lv.GetY(); s.lin_vel[2] = lv.GetZ();
				s.ang_vel[0] = av.GetX(); s.ang_vel[1] = av.GetY(); s.ang_vel[2] = av.GetZ();
				s.active = body_interface.IsActive(ids[i]) ? 1u : 0u;
				s.id = i;
			}
			const uint32_t st = (uint32_t)step;
			fwrite(&st, 4, 1, out);
			fwrite(states.data(), sizeof(sgp_body_state), n, out);
		}
	}
	fclose(out);
	printf("{\"steps\": %d, \"seconds\": %.6f, \"steps_per_s\": %.4f, \"threads\": %d, \"bodies\": %u, \"max_bodies\": %u, \"max_body_pairs\": %u, \"max_contact_constraints\": %u, \"jolt\": \"%s\"}\n",
		args.steps, seconds, seconds > 0 ? args.steps / seconds : 0.0, threads, n, cMaxBodies, cMaxBodyPairs, cMaxContactConstraints,
#ifdef JPH_VERSION_MAJOR
		(std::to_string(JPH_VERSION_MAJOR) + "." + std::to_string(JPH_VERSION_MINOR) + "." + std::to_string(JPH_VERSION_PATCH)).c_str()
#else
		"unknown"
#endif
	);
	for (uint32_t i = 0; i < n; ++i) { body_interface.RemoveBody(ids[i]); body_interface.DestroyBody(ids[i]); }
	JPH::UnregisterTypes();
	delete JPH::Factory::sInstance; JPH::Factory::sInstance = nullptr;
	return 0;
}
