// The receiving side of networked physics as GUIClient + ClientThread run it (docs/networked physics.txt): another client owns 32 dynamic
// boxes and streams an ObjectPhysicsTransformUpdate for each every 0.1 s of ITS clock; here the 80-byte records arrive late and jittered
// (ClientThread.cpp:736-792 -> ring of 4), and once per frame everything that is due (client_time + transmission_time_offset + 0.1 s,
// GUIClient.cpp:7462-7493) enters the physics world -- batched -- before PhysicsWorld::think().  The boxes then coast on the snapshot's
// velocities until the next one; smooth_translation hides the correction from the renderer.
#include "PhysicsWorld.h"
#include "PhysicsSnapshots.h"
#include <utils/Exception.h>
#include <cstdio>
#include <cmath>
#include <map>
#include <vector>
#include <algorithm>

int main()
{
	try {
		PhysicsWorld::init();
		Reference<PhysicsWorld> world = new PhysicsWorld(nullptr, nullptr);
		Reference<PhysicsObject> ground = new PhysicsObject(true, PhysicsWorld::createGroundQuadShape(2000.f), nullptr, 0);
		ground->pos = Vec4f(0, 0, -0.5f, 1);
		world->addObject(ground);
		const int N = 32;
		std::vector<Reference<PhysicsObject>> obs;
		std::map<uint64, PhysicsObject*> by_uid;
		for (int i = 0; i < N; ++i) {
			Reference<PhysicsObject> ob = new PhysicsObject(true);
			ob->is_cube = true; ob->scale = Vec3f(1.f, 1.f, 1.f); ob->mass = 50.f; ob->motion_type = PhysicsObject::MotionType_dynamic;
			ob->pos = Vec4f(3.f * (float)i, 0.f, 0.5f, 1.f);
			world->addObject(ob); world->activateObject(ob);
			obs.push_back(ob); by_uid[5000 + i] = ob.ptr();
		}
		PhysicsSnapshotQueue queue;
		const double skew = 12.25;                 // our global time minus the owner's
		for (int i = 0; i < N; ++i) queue.ownershipTaken(5000 + i, skew + 0.015, 0.0, /*renewal=*/false);      // the message took 15 ms
		// what the owner simulates: box i slides along +y at (1 + i / 16) m/s on the ground
		struct InFlight { double arrival; uint8_t msg[SGP_PHYSICS_UPDATE_BYTES]; };
		std::vector<InFlight> wire;
		unsigned rng = 12345u;
		for (int k = 1; k <= 30; ++k) {
			const double t_owner = 0.1 * k;
			for (int i = 0; i < N; ++i) {
				const float v = 1.f + (float)i / 16.f;
				InFlight f;
				writePhysicsTransformUpdate(5000 + i, Vec4f(3.f * (float)i, v * (float)t_owner, 0.5f, 1.f), Quatf::identity(), Vec4f(0, v, 0, 0), Vec4f(0, 0, 0, 0), t_owner, f.msg);
				rng = rng * 1664525u + 1013904223u;
				f.arrival = t_owner + skew + 0.02 + 0.06 * (double)(rng >> 8) / 16777216.0;      // 20 .. 80 ms
				wire.push_back(f);
			}
		}
		std::sort(wire.begin(), wire.end(), [](const InFlight& a, const InFlight& b) { return a.arrival < b.arrival; });
		size_t cursor = 0, inserted = 0, biggest = 0;
		double now = skew;
		float worst_jump = 0.f, worst = 0.f;
		for (int frame = 0; frame < 200; ++frame) {
			now += 1.0 / 60.0;
			while (cursor < wire.size() && wire[cursor].arrival <= now) { queue.receive(wire[cursor].msg, wire[cursor].arrival); ++cursor; }
			const size_t n = queue.insertDue(*world, now, [&](uint64_t uid) { auto it = by_uid.find(uid); return it == by_uid.end() ? (PhysicsObject*)nullptr : it->second; });
			inserted += n; biggest = std::max(biggest, n);
			// what the renderer shows straight after an insertion is what it showed before it (GUIClient.cpp:7484-7490)
			world->think(1.0 / 60.0);
			for (int i = 0; i < N; ++i) {
				// read-back as GUIClient.cpp:6581-6690 does, and decay of the smoothing offset
				JPH::RVec3 p; JPH::Quat q;
				world->physics_system->GetBodyInterface().GetPositionAndRotation(obs[i]->jolt_body_id, p, q);
				obs[i]->pos = Vec4f(p.GetX(), p.GetY(), p.GetZ(), 1.f);
				worst_jump = std::max(worst_jump, obs[i]->smooth_translation.length());
				obs[i]->smooth_translation = obs[i]->smooth_translation * 0.9f;
				// while the stream runs every box follows its owner: at our time `now` the owner's clock shows now - skew, and playback lags by
				// the 0.1 s padding (between snapshots the box coasts on the snapshot's velocity, minus a little ground friction)
				const double t_owner = now + 1.0 / 60.0 - skew - 0.1;
				if (t_owner > 0.5 && t_owner < 2.9) {
					const float v = 1.f + (float)i / 16.f;
					worst = std::max(worst, std::fabs(obs[i]->pos[1] - v * (float)t_owner));
				}
			}
		}
		printf("inserted %zu snapshots (largest batch %zu), worst |y - owner's y| %.3f m, largest smoothing offset %.3f m, tracked %u\n", inserted, biggest, worst, worst_jump, queue.expire(now));
		const bool ok = inserted == (size_t)(30 * N) && biggest >= 8 && worst < 0.25f && worst_jump < 0.5f;
		return ok ? 0 : 1;
	} catch (glare::Exception& e) { fprintf(stderr, "glare::Exception: %s\n", e.what().c_str()); return 2; }
}
