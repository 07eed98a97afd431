// See PhysicsObject.h.  Defaults: /root/reference/gui_client/PhysicsObject.cpp:25-60.
#include "PhysicsObject.h"
#include "PhysicsWorld.h"

js::AABBox PhysicsShape::getAABBOS() const
{
	Vec4f h(0.f);
	if (kind == 0) h = Vec4f(p[0], p[0], p[0], 0.f);
	else if (kind == 1) h = Vec4f(p[0], p[1], p[2], 0.f);
	else if (kind == 2) h = Vec4f(p[0], p[0], p[0] + p[1], 0.f);
	else if (kind == 4 && mesh && !mesh->vertices.empty()) {
		Vec4f mn(1e30f), mx(-1e30f);
		for (size_t i = 0; i + 2 < mesh->vertices.size(); i += 3) for (int k = 0; k < 3; ++k) { const float c = mesh->vertices[i + k]; if (c < mn[k]) mn[k] = c; if (c > mx[k]) mx[k] = c; }
		return js::AABBox(setWToOne(mn), setWToOne(mx));
	}
	else if (kind == 3 && hull && !hull->points.empty()) {
		Vec4f mn(1e30f), mx(-1e30f);
		for (size_t i = 0; i + 2 < hull->points.size(); i += 3) for (int k = 0; k < 3; ++k) { const float c = hull->points[i + k]; if (c < mn[k]) mn[k] = c; if (c > mx[k]) mx[k] = c; }
		return js::AABBox(setWToOne(mn), setWToOne(mx));
	}
	return js::AABBox(setWToOne(-h), setWToOne(h));
}

static void setDefaults(PhysicsObject& ob)
{
	ob.motion_type = PhysicsObject::MotionType_static;
	ob.is_sphere = false;
	ob.is_cube = false;
	ob.is_sensor = false;
	ob.mass = 100.f;
	ob.friction = 0.5f;
	ob.restitution = 0.3f;
	ob.use_zero_linear_drag = false;
	ob.underwater = false;
	ob.last_submerged_volume = 0;
	ob.scale = Vec3f(1.f);
	ob.body_com_os = Vec4f(0.f);
	ob.body_rot_os = Quatf::identity();
}

PhysicsObject::PhysicsObject(bool collidable_)
:	collidable(collidable_), userdata(NULL), userdata_type(0), pos(0.f), smooth_translation(0.f), smooth_rotation(Quatf::identity())
{
	setDefaults(*this);
}

PhysicsObject::PhysicsObject(bool collidable_, const PhysicsShape& shape_, void* userdata_, int userdata_type_)
:	shape(shape_), collidable(collidable_), userdata(userdata_), userdata_type(userdata_type_), pos(0.f), smooth_translation(0.f), smooth_rotation(Quatf::identity())
{
	setDefaults(*this);
}

PhysicsObject::~PhysicsObject() {}

const Matrix4f PhysicsObject::getObToWorldMatrix() const
{
	Matrix4f to_world, to_ob;
	computeToWorldAndToObMatrices(pos, rot, scale.toVec4fVector(), to_world, to_ob);
	return to_world;
}

const Matrix4f PhysicsObject::getWorldToObMatrix() const
{
	Matrix4f to_world, to_ob;
	computeToWorldAndToObMatrices(pos, rot, scale.toVec4fVector(), to_world, to_ob);
	return to_ob;
}

const js::AABBox PhysicsObject::getAABBoxWS() const
{
	// conservative: transform the 8 corners of the object-space box
	js::AABBox os = shape.getAABBOS();
	if (is_sphere) os = js::AABBox(Vec4f(-0.5f, -0.5f, -0.5f, 1), Vec4f(0.5f, 0.5f, 0.5f, 1));
	if (is_cube) os = js::AABBox(Vec4f(-0.5f, -0.5f, -0.5f, 1), Vec4f(0.5f, 0.5f, 0.5f, 1));
	const Matrix4f m = getObToWorldMatrix();
	Vec4f mn(1e30f), mx(-1e30f);
	for (int i = 0; i < 8; ++i) {
		const Vec4f c((i & 1) ? os.max_[0] : os.min_[0], (i & 2) ? os.max_[1] : os.min_[1], (i & 4) ? os.max_[2] : os.min_[2], 1.f);
		const Vec4f w = m * c;
		for (int k = 0; k < 3; ++k) { if (w[k] < mn[k]) mn[k] = w[k]; if (w[k] > mx[k]) mx[k] = w[k]; }
	}
	return js::AABBox(setWToOne(mn), setWToOne(mx));
}
