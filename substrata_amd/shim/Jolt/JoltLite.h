// JPH look-alike subset: exactly the Jolt types the physics FACADE's signatures and GUIClient's listeners name
// (PhysicsWorld.h:77-87,178-185; GUIClient.cpp:10588-10632).  Backed by the sgp C ABI, no Jolt code.
#pragma once
#include <cstdint>
#include <vector>
#include <cmath>
#include <string>
#include <unordered_map>
#include <memory>
struct PhysicsMeshData; struct PhysicsHullData;      // the facade's shape payloads (PhysicsObject.h)
namespace JPH
{
	typedef unsigned int uint;
	class Vec3
	{
	public:
		Vec3() : x(0), y(0), z(0) {}
		Vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
		float GetX() const { return x; } float GetY() const { return y; } float GetZ() const { return z; }
		Vec3 operator+(const Vec3& o) const { return Vec3(x + o.x, y + o.y, z + o.z); }
		Vec3 operator-(const Vec3& o) const { return Vec3(x - o.x, y - o.y, z - o.z); }
		Vec3 operator*(float f) const { return Vec3(x * f, y * f, z * f); }
		Vec3 operator/(float f) const { return Vec3(x / f, y / f, z / f); }
		Vec3 operator-() const { return Vec3(-x, -y, -z); }
		float Dot(const Vec3& o) const { return x * o.x + y * o.y + z * o.z; }
		Vec3 Cross(const Vec3& o) const { return Vec3(y * o.z - z * o.y, z * o.x - x * o.z, x * o.y - y * o.x); }
		float LengthSq() const { return x * x + y * y + z * z; }
		float Length() const { return std::sqrt(LengthSq()); }
		Vec3 Normalized() const { const float l = Length(); return Vec3(x / l, y / l, z / l); }
		Vec3 NormalizedOr(const Vec3& zero_value) const { const float l2 = LengthSq(); if (l2 <= 1.17549435e-38f) return zero_value; const float l = std::sqrt(l2); return Vec3(x / l, y / l, z / l); }
		bool IsNearZero(float max_dist_sq = 1.0e-12f) const { return LengthSq() <= max_dist_sq; }
		Vec3& operator+=(const Vec3& o) { x += o.x; y += o.y; z += o.z; return *this; }
		Vec3& operator-=(const Vec3& o) { x -= o.x; y -= o.y; z -= o.z; return *this; }
		Vec3& operator*=(float f) { x *= f; y *= f; z *= f; return *this; }
		float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
		static Vec3 sZero() { return Vec3(0, 0, 0); }
		static Vec3 sReplicate(float v) { return Vec3(v, v, v); }
		static Vec3 sAxisX() { return Vec3(1, 0, 0); }
		static Vec3 sAxisY() { return Vec3(0, 1, 0); }
		static Vec3 sAxisZ() { return Vec3(0, 0, 1); }
		float x, y, z;
	};
	inline Vec3 operator*(float f, const Vec3& v) { return Vec3(v.x * f, v.y * f, v.z * f); }
	struct Float3 { float x, y, z; };
	struct Float4 { float x, y, z, w; };
	// Jolt/Math/Math.h: the scalar helpers the controllers use (BikePhysics.cpp:442,939)
	static const float JPH_PI = 3.14159265358979323846f;
	inline float DegreesToRadians(float d) { return d * (JPH_PI / 180.0f); }
	inline float RadiansToDegrees(float r) { return r * (180.0f / JPH_PI); }
	template <class T> inline T Square(T v) { return v * v; }
	// the four floats a Quat is made from (JoltUtils.h:50 builds a Quat from one)
	class Vec4
	{
	public:
		Vec4() : x(0), y(0), z(0), w(0) {}
		Vec4(float x_, float y_, float z_, float w_) : x(x_), y(y_), z(z_), w(w_) {}
		float GetX() const { return x; } float GetY() const { return y; } float GetZ() const { return z; } float GetW() const { return w; }
		float x, y, z, w;
	};
	typedef Vec3 Vec3Arg_; // (Vec3Arg / RVec3Arg proper are declared in JoltCharacterLite.h)
	typedef Vec3 RVec3;
	class BodyID
	{
	public:
		static const uint32_t cInvalidBodyID = 0xFFFFFFFFu;
		BodyID() : id(cInvalidBodyID) {}
		explicit BodyID(uint32_t i) : id(i) {}
		bool IsInvalid() const { return id == cInvalidBodyID; }
		uint32_t GetIndexAndSequenceNumber() const { return id; }
		uint32_t GetIndex() const { return id; }
		bool operator==(const BodyID& o) const { return id == o.id; }
		bool operator!=(const BodyID& o) const { return id != o.id; }
	private:
		uint32_t id;
	};
	// Intrusive reference counting (JPH::RefTarget): a raw pointer taken out of a Ref and handed to another Ref shares the same count,
	// which is how CarPhysics passes shapes around (CarPhysics.cpp:71-78: Ref<Shape> -> const Shape* argument -> Ref<const Shape>).
	class RefTargetBase
	{
	public:
		RefTargetBase() : ref_count(0) {}
		RefTargetBase(const RefTargetBase&) : ref_count(0) {}
		RefTargetBase& operator=(const RefTargetBase&) { return *this; }
		virtual ~RefTargetBase() {}
		void AddRef() const { ++ref_count; }
		void Release() const { if (--ref_count == 0) delete this; }
		uint32_t GetRefCount() const { return ref_count; }
	private:
		mutable uint32_t ref_count;
	};
	class Quat
	{
	public:
		Quat() : x(0), y(0), z(0), w(1) {}
		Quat(float x_, float y_, float z_, float w_) : x(x_), y(y_), z(z_), w(w_) {}
		explicit Quat(const Vec4& v) : x(v.x), y(v.y), z(v.z), w(v.w) {}
		float GetX() const { return x; } float GetY() const { return y; } float GetZ() const { return z; } float GetW() const { return w; }
		Quat Conjugated() const { return Quat(-x, -y, -z, w); }
		static Quat sIdentity() { return Quat(0, 0, 0, 1); }
		// rotation of `angle` radians about the unit vector `axis` (Jolt: Quat::sRotation)
		static Quat sRotation(const Vec3& axis, float angle) { const float s = std::sin(0.5f * angle), c = std::cos(0.5f * angle); return Quat(axis.x * s, axis.y * s, axis.z * s, c); }
		Quat operator*(const Quat& o) const
		{
			return Quat(w * o.x + x * o.w + y * o.z - z * o.y, w * o.y - x * o.z + y * o.w + z * o.x, w * o.z + x * o.y - y * o.x + z * o.w, w * o.w - x * o.x - y * o.y - z * o.z);
		}
		Quat EnsureWPositive() const { return w < 0.0f ? Quat(-x, -y, -z, -w) : *this; }
		Quat Normalized() const { const float l = std::sqrt(x * x + y * y + z * z + w * w); return Quat(x / l, y / l, z / l, w / l); }
		Vec3 GetXYZ() const { return Vec3(x, y, z); }
		// Jolt: Quat::GetAxisAngle -- angle in [0, pi], axis zero for the identity
		void GetAxisAngle(Vec3& axis_out, float& angle_out) const
		{
			const Quat p = EnsureWPositive();
			if (p.w >= 1.0f) { axis_out = Vec3::sZero(); angle_out = 0.0f; }
			else { angle_out = 2.0f * std::acos(p.w); axis_out = p.GetXYZ().NormalizedOr(Vec3::sZero()); }
		}
		Vec3 operator*(const Vec3& v) const   // rotate
		{
			const float tx = 2 * (y * v.z - z * v.y), ty = 2 * (z * v.x - x * v.z), tz = 2 * (x * v.y - y * v.x);
			return Vec3(v.x + w * tx + (y * tz - z * ty), v.y + w * ty + (z * tx - x * tz), v.z + w * tz + (x * ty - y * tx));
		}
		float x, y, z, w;
	};
	// column-major rotation + translation, the part of JPH::Mat44 the controllers read
	class Mat44
	{
	public:
		Vec3 GetAxisX() const { return c[0]; } Vec3 GetAxisY() const { return c[1]; } Vec3 GetAxisZ() const { return c[2]; }
		Vec3 GetTranslation() const { return c[3]; }
		Vec3 GetColumn3(int i) const { return c[i]; }
		Vec3 operator*(const Vec3& v) const { return c[0] * v.x + c[1] * v.y + c[2] * v.z + c[3]; }
		Vec3 Multiply3x3(const Vec3& v) const { return c[0] * v.x + c[1] * v.y + c[2] * v.z; }
		// column-major float[16] with the implied bottom row (0, 0, 0, 1) (CarPhysics.cpp:331-334 feeds it to Matrix4f)
		void StoreFloat4x4(Float4* out) const { for (int i = 0; i < 4; ++i) { out[i].x = c[i].x; out[i].y = c[i].y; out[i].z = c[i].z; out[i].w = i == 3 ? 1.0f : 0.0f; } }
		static Mat44 sIdentity() { Mat44 m; m.c[0] = Vec3(1, 0, 0); m.c[1] = Vec3(0, 1, 0); m.c[2] = Vec3(0, 0, 1); m.c[3] = Vec3(0, 0, 0); return m; }
		static Mat44 sRotationTranslation(const Quat& q, const Vec3& t) { Mat44 m; m.c[0] = q * Vec3(1, 0, 0); m.c[1] = q * Vec3(0, 1, 0); m.c[2] = q * Vec3(0, 0, 1); m.c[3] = t; return m; }
		Mat44 operator*(const Mat44& o) const { Mat44 m; for (int i = 0; i < 3; ++i) m.c[i] = Multiply3x3(o.c[i]); m.c[3] = (*this) * o.c[3]; return m; }
		Vec3 c[4];
	};
	// JPH::SubShapeID: a path of child indices packed from the least significant bit (GUIClient.cpp:6485 pops one bit to tell the two
	// colliders of a portal's compound shape apart).  Here a compound's children are consecutive body slots, so the id of a contact on child k
	// of a compound is k in the lowest bits and all ones above (Jolt's "empty remainder").
	class SubShapeID
	{
	public:
		typedef uint32_t Type;
		SubShapeID() : value(0xFFFFFFFFu) {}
		explicit SubShapeID(uint32_t v) : value(v) {}
		uint32_t GetValue() const { return value; }
		void SetValue(uint32_t v) { value = v; }
		bool IsEmpty() const { return value == 0xFFFFFFFFu; }
		Type PopID(uint32_t bits, SubShapeID& remainder_out) const
		{
			const uint32_t mask = bits >= 32 ? 0xFFFFFFFFu : ((1u << bits) - 1u);
			const uint32_t fill = bits == 0 ? 0u : (mask << (32u - bits));
			const Type v = value & mask;
			remainder_out = SubShapeID(bits >= 32 ? 0xFFFFFFFFu : ((value >> bits) | fill));
			return v;
		}
		bool operator==(const SubShapeID& o) const { return value == o.value; }
		bool operator!=(const SubShapeID& o) const { return value != o.value; }
	private:
		uint32_t value;
	};
	enum class EMotionType : uint8_t { Static, Kinematic, Dynamic };
	enum class EOverrideMassProperties : uint8_t { CalculateMassAndInertia, CalculateInertia, MassAndInertiaProvided };
	struct MassProperties { float mMass = 0.0f; };
	template <class T> using Array = std::vector<T>;

	// JPH::Ref / JPH::RefConst over the intrusive count
	template <class T> class Ref
	{
	public:
		Ref() : ptr(nullptr) {}
		Ref(T* p) : ptr(p) { if (ptr) ptr->AddRef(); }
		Ref(const Ref& o) : ptr(o.ptr) { if (ptr) ptr->AddRef(); }
		template <class U> Ref(const Ref<U>& o) : ptr(o.GetPtr()) { if (ptr) ptr->AddRef(); }
		~Ref() { if (ptr) ptr->Release(); }
		Ref& operator=(const Ref& o) { if (o.ptr) o.ptr->AddRef(); if (ptr) ptr->Release(); ptr = o.ptr; return *this; }
		Ref& operator=(T* p) { if (p) p->AddRef(); if (ptr) ptr->Release(); ptr = p; return *this; }
		T* operator->() const { return ptr; }
		T& operator*() const { return *ptr; }
		T* GetPtr() const { return ptr; }
		operator T*() const { return ptr; }
		bool operator==(const T* p) const { return ptr == p; }
		bool operator!=(const T* p) const { return ptr != p; }
	private:
		T* ptr;
	};
	template <class T> using RefConst = Ref<const T>;

	// What a body needs from its shape.  kind follows SGP_SHAPE_*: 0 sphere (p0 = r), 1 box (p = half extents), 2 capsule (p0 = r,
	// p1 = half height of the cylinder, axis z), 3 convex hull (hull_points, com_offset), 4 triangle mesh, 5 static compound, -1 = only a volume (what TryGetBody reports).
	class Shape : public RefTargetBase
	{
	public:
		float GetVolume() const { return volume; }
		float volume = 0;
		int kind = -1;
		float p[3] = { 0, 0, 0 };
		std::vector<float> hull_points;          // xyz triples, shape space
		float com_offset[3] = { 0, 0, 0 };       // OffsetCenterOfMassShape
		// kind 4: a static triangle mesh as the facade's shape builders made it (PhysicsWorld::createMeshShape); kind 3 may carry the facade's hull data
		std::shared_ptr<PhysicsMeshData> mesh;
		std::shared_ptr<PhysicsHullData> hull;
		// kind 5: JPH::StaticCompoundShape -- children with their pose in the compound's space and Jolt's per-child user data
		struct SubShape { Vec3 pos; Quat rot; Ref<const Shape> shape; uint32_t user_data; };
		std::vector<SubShape> children;
		uint32_t GetNumSubShapes() const { return (uint32_t)children.size(); }
	};
	// What the listeners read from a body during a contact callback.
	class Body
	{
	public:
		const Shape* GetShape() const { return &shape; }
		Shape shape;
		// Where the simulated body frame (centre of mass, principal axes) sits in the object's shape space; identity except for convex
		// hulls.  Jolt hides this inside the body; PhysicsWorld::getJoltBody() fills it in and VehicleConstraint uses it to express the
		// wheel settings (given in shape space, like JPH::WheelSettings::mPosition) in the body frame.
		Vec3 com_offset; float frame_rot[4] = { 0, 0, 0, 1 };
		Vec3 GetLinearVelocity() const { return lin_vel; }
		uint64_t GetUserData() const { return user_data; }
		void SetUserData(uint64_t u) { user_data = u; }
		BodyID GetID() const { return id; }
		bool IsSensor() const { return is_sensor; }
		Vec3 lin_vel; uint64_t user_data = 0; BodyID id; bool is_sensor = false;
	};
	// JPH::StaticArray: fixed capacity, no heap (a ContactManifold is built per contact event, every step)
	template <class T, unsigned N> class StaticArray
	{
	public:
		typedef T value_type; typedef unsigned size_type;
		void push_back(const T& v) { if (n < N) new (begin() + n++) T(v); }
		void clear() { n = 0; }
		size_type size() const { return n; }
		bool empty() const { return n == 0; }
		T& operator[](size_type i) { return begin()[i]; }
		const T& operator[](size_type i) const { return begin()[i]; }
		T* begin() { return reinterpret_cast<T*>(buf); } T* end() { return begin() + n; }
		const T* begin() const { return reinterpret_cast<const T*>(buf); } const T* end() const { return begin() + n; }
	private:
		alignas(T) unsigned char buf[N * sizeof(T)];      // (raw storage: nothing is constructed until it is pushed; T is trivially destructible here)
		size_type n = 0;
	};
	class ContactManifold
	{
	public:
		RVec3 mBaseOffset;
		Vec3 mWorldSpaceNormal;
		float mPenetrationDepth = 0;
		typedef StaticArray<Vec3, 64> ContactPoints;      // (Jolt's capacity; a manifold here carries at most 4)
		ContactPoints mRelativeContactPointsOn1;
	};
	class ContactSettings {};


	// ---- shapes a caller builds itself (CarPhysics.cpp:66-78, BikePhysics.cpp:76-112) ------------------------------------------------
	class ShapeSettings : public RefTargetBase
	{
	public:
		// JPH::Result<Ref<Shape>>: CarPhysics only calls .Get()
		class ShapeResult
		{
		public:
			ShapeResult() {}
			explicit ShapeResult(Shape* s) : shape(s) {}
			explicit ShapeResult(const char* e) : error(e) {}
			bool IsValid() const { return shape.GetPtr() != nullptr; }
			bool HasError() const { return !IsValid(); }
			const std::string& GetError() const { return error; }
			Ref<Shape> Get() const { return shape; }
		private:
			Ref<Shape> shape; std::string error;
		};
		virtual ShapeResult Create() const = 0;
	};
	class SphereShapeSettings : public ShapeSettings
	{
	public:
		explicit SphereShapeSettings(float radius) : mRadius(radius) {}
		ShapeResult Create() const override { Shape* s = new Shape; s->kind = 0; s->p[0] = mRadius; s->volume = 4.18879020f * mRadius * mRadius * mRadius; return ShapeResult(s); }
		float mRadius;
	};
	class BoxShapeSettings : public ShapeSettings
	{
	public:
		explicit BoxShapeSettings(const Vec3& half_extent, float /*convex_radius*/ = 0.05f) : mHalfExtent(half_extent) {}
		ShapeResult Create() const override { Shape* s = new Shape; s->kind = 1; s->p[0] = mHalfExtent.x; s->p[1] = mHalfExtent.y; s->p[2] = mHalfExtent.z; s->volume = 8.0f * mHalfExtent.x * mHalfExtent.y * mHalfExtent.z; return ShapeResult(s); }
		Vec3 mHalfExtent;
	};
	class ConvexHullShapeSettings : public ShapeSettings
	{
	public:
		ConvexHullShapeSettings() {}
		explicit ConvexHullShapeSettings(const Array<Vec3>& points, float /*max_convex_radius*/ = 0.05f) : mPoints(points) {}
		ConvexHullShapeSettings(const Vec3* points, int n, float = 0.05f) : mPoints(points, points + n) {}
		ShapeResult Create() const override
		{
			if (mPoints.size() < 4) return ShapeResult("Too few points for a convex hull");          // (a flat cloud is reported when the body is created)
			Shape* s = new Shape; s->kind = 3;
			for (const Vec3& p : mPoints) { s->hull_points.push_back(p.x); s->hull_points.push_back(p.y); s->hull_points.push_back(p.z); }
			return ShapeResult(s);
		}
		Array<Vec3> mPoints;
	};
	// JPH::OffsetCenterOfMassShapeSettings(offset, inner): same collision geometry, centre of mass moved by `offset` (shape space)
	class OffsetCenterOfMassShapeSettings : public ShapeSettings
	{
	public:
		OffsetCenterOfMassShapeSettings(const Vec3& offset, const Shape* inner) : mOffset(offset), mInnerShapePtr(inner) {}
		ShapeResult Create() const override
		{
			if (!mInnerShapePtr.GetPtr() || mInnerShapePtr->kind != 3) return ShapeResult("OffsetCenterOfMassShape: implemented for convex hull shapes");
			Shape* s = new Shape(*mInnerShapePtr);
			s->com_offset[0] += mOffset.x; s->com_offset[1] += mOffset.y; s->com_offset[2] += mOffset.z;
			return ShapeResult(s);
		}
		Vec3 mOffset; RefConst<Shape> mInnerShapePtr;
	};

	// JPH::StaticCompoundShapeSettings (MeshBuilding.cpp:396-407): AddShape(position, rotation, shape or shape settings, user data) x n, Create()
	class StaticCompoundShapeSettings : public ShapeSettings
	{
	public:
		void AddShape(const Vec3& position, const Quat& rotation, const Shape* shape, uint32_t user_data = 0)
		{
			Shape::SubShape c; c.pos = position; c.rot = rotation; c.shape = shape; c.user_data = user_data; mSubShapes.push_back(c);
		}
		void AddShape(const Vec3& position, const Quat& rotation, const ShapeSettings* settings, uint32_t user_data = 0)
		{
			Ref<const ShapeSettings> keep(settings);
			const ShapeResult r = settings->Create();
			Shape::SubShape c; c.pos = position; c.rot = rotation; c.shape = r.Get().GetPtr(); c.user_data = user_data; mSubShapes.push_back(c);
		}
		ShapeResult Create() const override
		{
			if (mSubShapes.empty()) return ShapeResult("Compound needs a sub shape!");                     // Jolt's message
			for (const Shape::SubShape& c : mSubShapes) if (!c.shape.GetPtr() || c.shape->kind < 0 || c.shape->kind == 5) return ShapeResult("StaticCompoundShape: unsupported sub shape");
			Shape* s = new Shape; s->kind = 5; s->children = mSubShapes;
			return ShapeResult(s);
		}
		std::vector<Shape::SubShape> mSubShapes;
	};

	// JPH::BodyCreationSettings: the fields CarPhysics / BikePhysics set (CarPhysics.cpp:80-84) plus Jolt's defaults for the rest
	class BodyCreationSettings
	{
	public:
		BodyCreationSettings() {}
		BodyCreationSettings(const Shape* shape, const RVec3& position, const Quat& rotation, EMotionType motion_type, uint16_t object_layer)
			: mPosition(position), mRotation(rotation), mMotionType(motion_type), mObjectLayer(object_layer), mShape(shape) {}
		const Shape* GetShape() const { return mShape; }
		void SetShape(const Shape* s) { mShape = s; }
		RVec3 mPosition; Quat mRotation; Vec3 mLinearVelocity, mAngularVelocity;
		uint64_t mUserData = 0;
		EMotionType mMotionType = EMotionType::Dynamic;
		uint16_t mObjectLayer = 0;
		bool mIsSensor = false, mAllowSleeping = true;
		float mFriction = 0.2f, mRestitution = 0.0f, mLinearDamping = 0.05f, mAngularDamping = 0.05f, mGravityFactor = 1.0f;
		EOverrideMassProperties mOverrideMassProperties = EOverrideMassProperties::CalculateMassAndInertia;
		MassProperties mMassPropertiesOverride;
	private:
		RefConst<Shape> mShape;
	};
	enum class EActivation { Activate, DontActivate };
}

struct sgp_world;

namespace JPH
{
	// JPH::BodyInterface look-alike: the calls HoverCarPhysics.cpp:113-348,425-480, BoatPhysics.cpp:35-49,134-267,367-385 and
	// GUIClient.cpp:6577-6673 make through physics_world.physics_system->GetBodyInterface(), forwarded to the sgp C ABI.
	// Getters share one cached read-back per body between world mutations (invalidate() is called by PhysicsWorld::think and setters).
	class BodyInterface
	{
	public:
		explicit BodyInterface(sgp_world* w) : world(w), cached_id(0xFFFFFFFFu) {}
		~BodyInterface();
		// CarPhysics.cpp:84-90: the body is built here (hull shapes become a device hull; the body frame = centre of mass / principal axes
		// is remembered so that every getter below answers in the SHAPE's space, like Jolt) and enters the world asleep; AddBody wakes it.
		// Returns nullptr when the world refuses the body (capacity, degenerate hull), like Jolt when it runs out of bodies.
		Body* CreateBody(const BodyCreationSettings& settings);
		void AddBody(const BodyID& id, EActivation activation);
		BodyID CreateAndAddBody(const BodyCreationSettings& settings, EActivation activation) { Body* b = CreateBody(settings); if (!b) return BodyID(); AddBody(b->GetID(), activation); return b->GetID(); }
		void RemoveBody(const BodyID& id);
		void DestroyBody(const BodyID& id);
		bool IsAdded(const BodyID& id) const { return bodies.count(id.GetIndex()) != 0; }
		// where the simulated body frame of `id` sits in its shape's space (identity unless the body is a convex hull)
		struct Frame { Vec3 com; Quat rot; };
		void setFrame(const BodyID& id, const Vec3& com, const Quat& rot);
		void clearFrame(const BodyID& id);
		const Frame* getFrame(const BodyID& id) const;
		const Body* findBody(const BodyID& id) const { auto it = bodies.find(id.GetIndex()); return it == bodies.end() ? nullptr : it->second; }
		void ActivateBody(const BodyID& id);
		void DeactivateBody(const BodyID&) {}
		void AddForce(const BodyID& id, const Vec3& force);
		void AddForce(const BodyID& id, const Vec3& force, const RVec3& point);
		void AddTorque(const BodyID& id, const Vec3& torque);
		RVec3 GetPosition(const BodyID& id) const;
		RVec3 GetCenterOfMassPosition(const BodyID& id) const;
		Quat GetRotation(const BodyID& id) const;
		void GetPositionAndRotation(const BodyID& id, RVec3& pos_out, Quat& rot_out) const;
		Mat44 GetWorldTransform(const BodyID& id) const;
		Vec3 GetLinearVelocity(const BodyID& id) const;
		Vec3 GetAngularVelocity(const BodyID& id) const;
		void GetLinearAndAngularVelocity(const BodyID& id, Vec3& lin_out, Vec3& ang_out) const;
		Vec3 GetPointVelocity(const BodyID& id, const RVec3& point) const;
		void SetLinearAndAngularVelocity(const BodyID& id, const Vec3& lin, const Vec3& ang);
		bool IsActive(const BodyID& id) const;
		void invalidate() const { cached_id = 0xFFFFFFFFu; }
	private:
		void fetch(const BodyID& id) const;
		sgp_world* world;
		mutable uint32_t cached_id;
		mutable float st_pos[3], st_rot[4], st_lv[3], st_av[3];
		mutable bool st_active;
		std::unordered_map<uint32_t, Body*> bodies;       // bodies made by CreateBody
		std::unordered_map<uint32_t, Frame> frames;
	};

	// GetBodyLockInterface().TryGetBody(id)->GetShape()->GetVolume()  (BoatPhysics.cpp:40-43)
	class BodyLockInterface
	{
	public:
		explicit BodyLockInterface(sgp_world* w) : world(w) {}
		Body* TryGetBody(const BodyID& id) const;          // valid until the next TryGetBody call
		bool fill(const BodyID& id, Body& out) const;
	private:
		sgp_world* world; mutable Body scratch;
	};

	// JPH::BodyLockRead (PlayerPhysics.cpp:519-545): the body's user data is what the caller is after
	class BodyLockRead
	{
	public:
		BodyLockRead(const BodyLockInterface& iface, const BodyID& id);
		bool Succeeded() const { return ok; }
		bool SucceededAndIsInBroadPhase() const { return ok; }
		const Body& GetBody() const { return body; }
		void ReleaseLock() {}
	private:
		Body body; bool ok;
	};
	typedef BodyLockRead BodyLockWrite;

	class BroadPhaseLayerFilter {};
	class ObjectLayerFilter { public: virtual ~ObjectLayerFilter() {} virtual bool ShouldCollide(uint16_t) const { return true; } };
	class DefaultBroadPhaseLayerFilter : public BroadPhaseLayerFilter {};
	class DefaultObjectLayerFilter : public ObjectLayerFilter {};

	class VehicleConstraint;      // Jolt/JoltVehicleLite.h
	class PhysicsStepListener { public: virtual ~PhysicsStepListener() {} };

	class PhysicsSystem
	{
	public:
		explicit PhysicsSystem(sgp_world* w) : world(w), body_interface(w), body_lock_interface(w), step_serial(0) {}
		const BodyLockInterface& GetBodyLockInterface() const { return body_lock_interface; }
		BodyInterface& GetBodyInterface() { return body_interface; }
		const BodyInterface& GetBodyInterface() const { return body_interface; }
		Vec3 GetGravity() const { return Vec3(0, 0, -9.81f); }   // PhysicsWorld.cpp:520
		// CarPhysics.cpp:224-226,258-262: the vehicle constraint is both a constraint and a step listener in Jolt; here the
		// constraint registration creates / destroys the device-side vehicle and the listener calls are no-ops.
		void AddConstraint(VehicleConstraint* c);
		void RemoveConstraint(VehicleConstraint* c);
		void AddStepListener(PhysicsStepListener*) {}
		void RemoveStepListener(PhysicsStepListener*) {}
		void onStep() { ++step_serial; body_interface.invalidate(); }       // called by PhysicsWorld::think
		// compound bodies (StaticCompoundShape): how many children body `id` has, and the JPH::SubShapeID of child k -- k in the lowest
		// ceil(log2(n)) bits, ones above (Jolt's empty remainder), so that SubShapeID::PopID(bits, remainder) returns k (GUIClient.cpp:6484-6486)
		void registerCompound(const BodyID& id, uint32_t num_children) { if (num_children) compound_sizes[id.GetIndex()] = num_children; else compound_sizes.erase(id.GetIndex()); }
		SubShapeID subShapeID(const BodyID& id, uint32_t child) const
		{
			auto it = compound_sizes.find(id.GetIndex());
			if (it == compound_sizes.end()) return SubShapeID();
			uint32_t bits = 0; while ((1u << bits) < it->second) ++bits;
			if (bits == 0) bits = 1;                                        // (Jolt uses at least one bit for a compound)
			return SubShapeID(bits >= 32 ? child : ((0xFFFFFFFFu << bits) | child));
		}
		// filter factories CharacterVirtual callers pass through (PlayerPhysics.cpp:106-114,344-346): the character queries always use the
		// MOVING object layer's collision set, so these are placeholders
		DefaultBroadPhaseLayerFilter GetDefaultBroadPhaseLayerFilter(uint16_t) const { return DefaultBroadPhaseLayerFilter(); }
		DefaultObjectLayerFilter GetDefaultLayerFilter(uint16_t) const { return DefaultObjectLayerFilter(); }
		sgp_world* world;
	private:
		BodyInterface body_interface;
		BodyLockInterface body_lock_interface;
		uint64_t step_serial;
		std::unordered_map<uint32_t, uint32_t> compound_sizes;
	};
}
