// A portal as the reference builds and uses it:
//   MeshBuilding::makePortalMeshes (gui_client/MeshBuilding.cpp:377-413): the arch mesh with create_tris_for_mat[3] = false (the blue
//     portal plane is not collidable) + a thin box across the opening, as a JPH::StaticCompoundShape with per-child user data
//   GUIClient (GUIClient.cpp:2379-2393): a static, collidable PhysicsObject carrying that shape, rotated and scaled like any object
//   PlayerPhysics::OnContactAdded (PlayerPhysics.cpp:519-533): BodyLockRead -> user data -> contacted_events {object, sub shape id, position}
//   GUIClient (GUIClient.cpp:6482-6491): contacted_events[z].sub_shape_id.PopID(1, remainder) == 1  <=>  the inner plane collider was touched
// The statements are the reference's; portal.bmesh (absent from the tree) is replaced by a synthetic arch with the same four materials.
#include "PhysicsWorld.h"
#include "JoltUtils.h"
#include <utils/Exception.h>
#include <Jolt/Jolt.h>
#include <Jolt/Physics/Collision/ObjectLayer.h>
#include <Jolt/Physics/Character/Character.h>
#include <Jolt/Physics/Character/CharacterVirtual.h>
#include <Jolt/Physics/PhysicsSystem.h>
#include <Jolt/Physics/Collision/Shape/CapsuleShape.h>
#include <Jolt/Physics/Collision/Shape/RotatedTranslatedShape.h>
#include <cstdio>
#include <cmath>
#include <vector>

static void addBoxTris(std::vector<Vec3f>& v, std::vector<uint32>& t, std::vector<uint32>& m, const Vec3f& lo, const Vec3f& hi, uint32 mat)
{
	const uint32 b = (uint32)v.size();
	for (int i = 0; i < 8; ++i) v.push_back(Vec3f((i & 1) ? hi.x : lo.x, (i & 2) ? hi.y : lo.y, (i & 4) ? hi.z : lo.z));
	const uint32 quads[6][4] = { { 0, 2, 3, 1 }, { 4, 5, 7, 6 }, { 0, 1, 5, 4 }, { 2, 6, 7, 3 }, { 0, 4, 6, 2 }, { 1, 3, 7, 5 } };      // outward-facing
	for (int q = 0; q < 6; ++q) { t.push_back(b + quads[q][0]); t.push_back(b + quads[q][1]); t.push_back(b + quads[q][2]); t.push_back(b + quads[q][0]); t.push_back(b + quads[q][2]); t.push_back(b + quads[q][3]); m.push_back(mat); m.push_back(mat); }
}

struct ContactedEvent { PhysicsObject* ob; JPH::SubShapeID sub_shape_id; JPH::Vec3 pos; };

struct PlayerPhysicsLike : public JPH::CharacterContactListener
{
	JPH::PhysicsSystem* physics_system = nullptr;
	std::vector<ContactedEvent> contacted_events;
	JPH::CharRef<JPH::CharacterShape> standing_shape;
	std::unique_ptr<JPH::CharacterVirtual> jolt_character;
	JPH::TempAllocator temp_allocator;

	// PlayerPhysics.cpp:519-533
	void OnContactAdded(const JPH::CharacterVirtual* inCharacter, const JPH::BodyID& inBodyID2, const JPH::SubShapeID& inSubShapeID2, JPH::RVec3Arg inContactPosition, JPH::Vec3Arg inContactNormal, JPH::CharacterContactSettings& ioSettings) override
	{
		JPH::BodyLockRead lock(physics_system->GetBodyLockInterface(), inBodyID2);
		if(lock.Succeeded())
		{
			const JPH::Body& body = lock.GetBody();
			const uint64 user_data = body.GetUserData();
			if(user_data != 0)
			{
				PhysicsObject* physics_ob = (PhysicsObject*)user_data;
				contacted_events.push_back(ContactedEvent({physics_ob, inSubShapeID2, inContactPosition}));
			}
		}
	}
	void init(PhysicsWorld& physics_world, const JPH::Vec3& bottom_pos)
	{
		physics_system = physics_world.physics_system;
		standing_shape = JPH::RotatedTranslatedShapeSettings(JPH::Vec3(0, 0, 0.65f + 0.3f), JPH::Quat(0.7071068f, 0, 0, 0.7071068f), new JPH::CapsuleShape(0.65f, 0.3f)).Create().Get();
		JPH::CharRef<JPH::CharacterVirtualSettings> settings = new JPH::CharacterVirtualSettings();
		settings->mShape = standing_shape; settings->mUp = JPH::Vec3(0, 0, 1); settings->mSupportingVolume = JPH::Plane(JPH::Vec3(0, 0, 1), -0.3f); settings->mMaxStrength = 1000;
		jolt_character.reset(new JPH::CharacterVirtual(settings, bottom_pos, JPH::Quat(), physics_world.physics_system));
		jolt_character->SetListener(this);
	}
	void update(PhysicsWorld& physics_world, const JPH::Vec3& move_desired_vel, float dtime)
	{
		JPH::Vec3 vel = jolt_character->GetLinearVelocity();
		jolt_character->UpdateGroundVelocity();
		if (jolt_character->IsSupported()) vel = move_desired_vel + jolt_character->GetGroundVelocity(); else vel = vel + move_desired_vel * dtime;
		vel = vel + JPH::Vec3(0, 0, -9.81f) * dtime;
		jolt_character->SetLinearVelocity(vel);
		JPH::CharacterVirtual::ExtendedUpdateSettings settings;
		settings.mStickToFloorStepDown = JPH::Vec3(0, 0, -0.5f); settings.mWalkStairsStepUp = JPH::Vec3(0.0f, 0.0f, 0.4f);
		jolt_character->ExtendedUpdate(dtime, physics_world.physics_system->GetGravity(), settings, physics_world.physics_system->GetDefaultBroadPhaseLayerFilter(1),
			physics_world.physics_system->GetDefaultLayerFilter(1), JPH::BodyFilter(), JPH::ShapeFilter(), temp_allocator);
	}
};

#define CHECK(cond) do { if (!(cond)) { std::printf("FAILED: %s (line %d)\n", #cond, __LINE__); return 1; } } while (0)

int main()
{
	try {
		PhysicsWorld::init();
		Reference<PhysicsWorld> world = new PhysicsWorld(nullptr, nullptr);
		Reference<PhysicsObject> ground = new PhysicsObject(true, PhysicsWorld::createGroundQuadShape(2000.f), nullptr, 0);
		ground->pos = Vec4f(0, 0, -0.5f, 1);
		world->addObject(ground);

		// ------------------------------------------------------------------------- MeshBuilding::makePortalMeshes, :377-413
		// (portal.bmesh: two posts + a lintel = material 1 "arch", the inside rim = material 0, the blue portal plane = material 3)
		std::vector<Vec3f> verts; std::vector<uint32> tris, tri_mats;
		addBoxTris(verts, tris, tri_mats, Vec3f(-0.9f, -0.15f, 0.f), Vec3f(-0.55f, 0.15f, 2.2f), 1);
		addBoxTris(verts, tris, tri_mats, Vec3f(0.55f, -0.15f, 0.f), Vec3f(0.9f, 0.15f, 2.2f), 1);
		addBoxTris(verts, tris, tri_mats, Vec3f(-0.9f, -0.15f, 2.2f), Vec3f(0.9f, 0.15f, 2.6f), 1);
		addBoxTris(verts, tris, tri_mats, Vec3f(-0.55f, -0.01f, 0.f), Vec3f(0.55f, 0.01f, 2.2f), 3);          // the portal plane, both faces

		std::vector<bool> create_tris_for_mat(4, true);
		create_tris_for_mat[3] = false; // Material with index 3 is the blue portal shader material that shouldn't be collidable.
		PhysicsShape arch_shape = PhysicsWorld::createMeshShape(verts, tris, &tri_mats, &create_tris_for_mat);      // (createJoltShapeForBatchedMesh(*batched_mesh, false, nullptr, &create_tris_for_mat))

		JPH::Ref<JPH::StaticCompoundShapeSettings> compound_settings = new JPH::StaticCompoundShapeSettings();
		compound_settings->AddShape(JPH::Vec3Arg(0,0,0), JPH::QuatArg::sIdentity(), arch_shape.jolt_shape, /*inUserData=*/0);

		JPH::Ref<JPH::BoxShapeSettings> box_settings = new JPH::BoxShapeSettings(/*inHalfExtent=*/JPH::Vec3Arg(0.5f, 0.06f, 1.f));

		compound_settings->AddShape(/*position=*/JPH::Vec3Arg(0,0,1.f), JPH::QuatArg::sIdentity(), box_settings, /*inUserData=*/1);

		JPH::ShapeSettings::ShapeResult result = compound_settings->Create();
		if(result.HasError())
			throw glare::Exception(std::string("Error building Jolt shape: ") + result.GetError().c_str());
		JPH::Ref<JPH::Shape> compound_shape = result.Get();

		PhysicsShape portal_shape;
		portal_shape.jolt_shape = compound_shape;
		portal_shape.size_B = PhysicsWorld::computeSizeBForShape(compound_shape);
		CHECK(portal_shape.size_B > sizeof(JPH::Shape) && compound_shape->GetNumSubShapes() == 2);

		// ------------------------------------------------------------------------- GUIClient.cpp:2379-2393
		int world_object_stand_in = 0;
		PhysicsObjectRef physics_ob = new PhysicsObject(/*collidable=*/true);
		physics_ob->shape = portal_shape;
		physics_ob->is_sensor = false;
		physics_ob->userdata = &world_object_stand_in;
		physics_ob->userdata_type = 0;
		physics_ob->pos = Vec4f(6.f, 0.f, 0.f, 1);
		physics_ob->rot = Quatf::fromAxisAndAngle(normalise(Vec4f(0, 0, 1, 0)), 1.5707963f);          // the opening faces +-x
		physics_ob->scale = Vec3f(1.f);
		world->addObject(physics_ob);
		CHECK(!physics_ob->jolt_body_id.IsInvalid() && world->getNumObjects() == 2);

		// rays through the facade: the (filtered) portal plane is not there, the inner box is; the post reports its material
		RayTraceResult r;
		world->traceRay(Vec4f(0, 0, 1.2f, 1), Vec4f(1, 0, 0, 0), 100.f, JPH::BodyID(), r);
		CHECK(r.hit_object == physics_ob.ptr() && std::fabs(r.hit_t - (6.f - 0.06f)) < 1e-3f && r.hit_mat_index == 0);       // the box (no triangle -> 0)
		world->traceRay(Vec4f(0, 0.7f, 1.2f, 1), Vec4f(1, 0, 0, 0), 100.f, JPH::BodyID(), r);
		CHECK(r.hit_object == physics_ob.ptr() && std::fabs(r.hit_t - (6.f - 0.15f)) < 1e-3f && r.hit_mat_index == 1);       // a post: the arch material

		// ------------------------------------------------------------------------- the player walks into the opening, later into a post
		PlayerPhysicsLike player_physics;
		player_physics.init(*world, JPH::Vec3(2.f, 0.f, 0.5f));
		std::string touched_portal;
		int touched_arch = 0;
		auto frames = [&](int n, const JPH::Vec3& desired) {
			for (int s = 0; s < n; ++s) {
				player_physics.update(*world, desired, 1.f / 60.f);
				world->think(1.0 / 60.0);
				// GUIClient.cpp:6470-6494
				for (size_t z = 0; z < player_physics.contacted_events.size(); ++z) {
					PhysicsObject* ob = player_physics.contacted_events[z].ob;
					if (ob == physics_ob.ptr())      // (ob->object_type == WorldObject::ObjectType_Portal)
					{
						JPH::SubShapeID remainder;
						const uint32 subshape = player_physics.contacted_events[z].sub_shape_id.PopID(/*num bits=*/1, remainder);
						if(subshape == 1) // If touched inner plane collider in portal:
							touched_portal = "target_url";
						else ++touched_arch;
					}
				}
				player_physics.contacted_events.resize(0);
			}
		};
		frames(60, JPH::Vec3(0, 0, 0));
		CHECK(touched_portal.empty() && touched_arch == 0);
		frames(150, JPH::Vec3(3, 0, 0));                                     // straight at the opening
		const JPH::Vec3 p1 = player_physics.jolt_character->GetPosition();
		std::printf("at the opening: %.3f %.3f %.3f  touched portal '%s'  arch contacts %d\n", p1.x, p1.y, p1.z, touched_portal.c_str(), touched_arch);
		CHECK(touched_portal == "target_url" && touched_arch == 0);
		CHECK(std::fabs(p1.x - (6.f - 0.06f - 0.3f)) < 0.08f);               // stopped by the inner collider (it is collidable, GUIClient.cpp:2384)
		// walk away, then against the +y post (world y = +0.55 .. 0.9 after the quarter turn): the arch child, not the portal
		touched_portal.clear();
		player_physics.jolt_character->SetPosition(JPH::Vec3(3.f, 0.72f, 0.f)); player_physics.jolt_character->SetLinearVelocity(JPH::Vec3(0, 0, 0));
		frames(150, JPH::Vec3(3, 0, 0));
		const JPH::Vec3 p2 = player_physics.jolt_character->GetPosition();
		std::printf("at the post:    %.3f %.3f %.3f  touched portal '%s'  arch contacts %d\n", p2.x, p2.y, p2.z, touched_portal.c_str(), touched_arch);
		CHECK(touched_portal.empty() && touched_arch >= 1 && p2.x < 6.f - 0.15f - 0.3f + 0.08f);

		// the portal is carried elsewhere, scaled up, and removed like any object
		world->setNewObToWorldTransform(*physics_ob, Vec4f(-10.f, 4.f, 0.f, 1), Quatf::identity(), Vec4f(2.f, 2.f, 2.f, 0));
		world->traceRay(Vec4f(-10.f, -5.f, 2.4f, 1), Vec4f(0, 1, 0, 0), 100.f, JPH::BodyID(), r);
		CHECK(r.hit_object == physics_ob.ptr() && std::fabs(r.hit_t - (9.f - 0.12f)) < 1e-3f);             // the box, twice as thick
		world->traceRay(Vec4f(0, 0, 1.2f, 1), Vec4f(1, 0, 0, 0), 100.f, JPH::BodyID(), r);
		CHECK(r.hit_object != physics_ob.ptr());
		world->removeObject(physics_ob);
		CHECK(world->getNumObjects() == 1);
		std::printf("portal_walkthrough: ok\n");
		return 0;
	} catch (glare::Exception& e) { std::printf("exception: %s\n", e.what().c_str()); return 2; }
}
