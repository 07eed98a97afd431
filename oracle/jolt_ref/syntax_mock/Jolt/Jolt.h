// SYNTAX-CHECK MOCK, NOT JOLT.  Declarations (no definitions) of exactly the JoltPhysics v5.3.0 names oracle_jolt.cpp uses, written from the
// way that file uses them, so that `g++ -fsyntax-only -I oracle/jolt_ref/syntax_mock oracle/jolt_ref/oracle_jolt.cpp` keeps the driver from
// rotting while Jolt's sources are absent (tests/test_jolt_ref.py::test_oracle_jolt_driver_still_parses).  Nothing here can be linked or run,
// nothing is built from it, and it is never on an include path of the product or of oracle/_ref.  With the real Jolt: make -C oracle jolt_ref
// SGP_JOLT_DIR=<JoltPhysics checkout>.
#pragma once
#include <cstdint>
#include <cstddef>
#include <vector>
#define JPH_SUPPRESS_WARNINGS
namespace JPH
{
	typedef unsigned int uint;
	typedef uint16_t ObjectLayer;
	static constexpr float JPH_PI = 3.14159265358979323846f;
	static constexpr uint cMaxPhysicsJobs = 2048, cMaxPhysicsBarriers = 8;
	class BroadPhaseLayer { public: constexpr BroadPhaseLayer() : v(0xFF) {} explicit constexpr BroadPhaseLayer(uint8_t x) : v(x) {} constexpr bool operator==(const BroadPhaseLayer& o) const { return v == o.v; } private: uint8_t v; };
	class Vec3 { public: Vec3(); Vec3(float, float, float); float GetX() const; float GetY() const; float GetZ() const; bool operator==(const Vec3&) const; static Vec3 sZero(); static Vec3 sAxisX(); };
	typedef Vec3 RVec3;
	class Quat { public: Quat(); Quat(float, float, float, float); float GetX() const; float GetY() const; float GetZ() const; float GetW() const; static Quat sRotation(const Vec3&, float); };
	struct AABox { Vec3 mMin, mMax; };
	template <class T> class Ref { public: Ref(); Ref(T*); template <class U> Ref(const Ref<U>&); template <class U> Ref& operator=(const Ref<U>&); Ref& operator=(T*); T* operator->() const; };
	class BodyID { public: BodyID(); bool IsInvalid() const; };
	typedef std::vector<BodyID> BodyIDVector;
	enum class EMotionType : uint8_t { Static, Kinematic, Dynamic };
	enum class EActivation { Activate, DontActivate };
	enum class EBodyType : uint8_t { RigidBody, SoftBody };
	enum class EOverrideMassProperties : uint8_t { CalculateMassAndInertia, CalculateInertia, MassAndInertiaProvided };
	struct MassProperties { float mMass; };
	class ShapeSettings { public: virtual ~ShapeSettings(); };
	class SphereShapeSettings : public ShapeSettings { public: explicit SphereShapeSettings(float); };
	class BoxShapeSettings : public ShapeSettings { public: explicit BoxShapeSettings(const Vec3&); };
	class CapsuleShapeSettings : public ShapeSettings { public: CapsuleShapeSettings(float, float); };
	class ScaledShapeSettings : public ShapeSettings { public: ScaledShapeSettings(const ShapeSettings*, const Vec3&); template <class T> ScaledShapeSettings(const Ref<T>&, const Vec3&); };
	class RotatedTranslatedShapeSettings : public ShapeSettings { public: template <class T> RotatedTranslatedShapeSettings(const Vec3&, const Quat&, const Ref<T>&); };
	class Shape { public: float GetVolume() const; };
	class MotionProperties { public: float GetInverseMass() const; };
	class Body
	{
	public:
		EMotionType GetMotionType() const; const AABox& GetWorldSpaceBounds() const; const Shape* GetShape() const; const MotionProperties* GetMotionProperties() const;
		bool ApplyBuoyancyImpulse(const RVec3&, const Vec3&, float, float, float, const Vec3&, const Vec3&, float);
	};
	class BodyCreationSettings
	{
	public:
		template <class T> BodyCreationSettings(const Ref<T>&, const RVec3&, const Quat&, EMotionType, ObjectLayer);
		bool mIsSensor, mAllowSleeping; float mFriction, mRestitution, mGravityFactor, mLinearDamping, mAngularDamping;
		MassProperties mMassPropertiesOverride; EOverrideMassProperties mOverrideMassProperties; Vec3 mLinearVelocity, mAngularVelocity; uint64_t mUserData;
	};
	class BodyInterface
	{
	public:
		BodyID CreateAndAddBody(const BodyCreationSettings&, EActivation); void ActivateBody(const BodyID&); bool IsActive(const BodyID&) const;
		void GetPositionAndRotation(const BodyID&, RVec3&, Quat&) const; void GetLinearAndAngularVelocity(const BodyID&, Vec3&, Vec3&) const;
		void RemoveBody(const BodyID&); void DestroyBody(const BodyID&);
	};
	class BodyLockInterface {};
	class BodyLockWrite { public: BodyLockWrite(const BodyLockInterface&, const BodyID&); bool Succeeded() const; Body& GetBody() const; };
	class BroadPhaseLayerInterface { public: virtual ~BroadPhaseLayerInterface(); virtual uint GetNumBroadPhaseLayers() const = 0; virtual BroadPhaseLayer GetBroadPhaseLayer(ObjectLayer) const = 0; };
	class ObjectVsBroadPhaseLayerFilter { public: virtual ~ObjectVsBroadPhaseLayerFilter(); virtual bool ShouldCollide(ObjectLayer, BroadPhaseLayer) const; };
	class ObjectLayerPairFilter { public: virtual ~ObjectLayerPairFilter(); virtual bool ShouldCollide(ObjectLayer, ObjectLayer) const; };
	class TempAllocator { public: virtual ~TempAllocator(); };
	class TempAllocatorMalloc : public TempAllocator {};
	class JobSystem { public: virtual ~JobSystem(); };
	class JobSystemThreadPool : public JobSystem { public: JobSystemThreadPool(uint, uint, int); };
	class PhysicsSystem
	{
	public:
		void Init(uint, uint, uint, uint, const BroadPhaseLayerInterface&, const ObjectVsBroadPhaseLayerFilter&, const ObjectLayerPairFilter&);
		void SetGravity(const Vec3&); BodyInterface& GetBodyInterface(); void OptimizeBroadPhase();
		int Update(float, int, TempAllocator*, JobSystem*); void GetActiveBodies(EBodyType, BodyIDVector&) const; const BodyLockInterface& GetBodyLockInterface() const;
	};
	class Factory { public: static Factory* sInstance; };
	void RegisterDefaultAllocator(); void RegisterTypes(); void UnregisterTypes();
}
