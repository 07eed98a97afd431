"""The C++ drop-in facade (substrata_amd/shim: PhysicsWorld / PhysicsObject with the reference's names) driven like
GUIClient drives it, compared with the oracle driven through the C-level calls on the same scene (BASELINE config 1)."""
import os
import subprocess

import numpy as np
import pytest

from substrata_amd import abi, scenes, build, build_shim
from helpers import DT

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_facade_exe(tmp_path, src="facade_scene.cpp"):
    build.build()
    build_shim.build()
    exe = str(tmp_path / src.replace(".cpp", ""))
    lib_dir = os.path.join(ROOT, "substrata_amd")
    cmd = ["g++", "-O2", "-std=c++17", "-I", os.path.join(lib_dir, "shim"), os.path.join(ROOT, "tests", "cpp", src),
           "-o", exe, "-L", lib_dir, "-lsgp_shim", "-lsgp", f"-Wl,-rpath,{lib_dir}"]
    subprocess.run(cmd, check=True)
    return exe


def test_facade_compiles_without_gpu(tmp_path):
    """CPU check: the facade and a GUIClient-style caller compile and link against libsgp.so (no run)."""
    assert os.path.exists(build_facade_exe(tmp_path))
    assert os.path.exists(build_facade_exe(tmp_path, "hover_controller.cpp"))
    assert os.path.exists(build_facade_exe(tmp_path, "car_controller.cpp"))
    assert os.path.exists(build_facade_exe(tmp_path, "car_physics_sequence.cpp"))
    assert os.path.exists(build_facade_exe(tmp_path, "portal_walkthrough.cpp"))
    assert os.path.exists(build_facade_exe(tmp_path, "bike_controller.cpp"))
    assert os.path.exists(build_facade_exe(tmp_path, "player_controller.cpp"))
    assert os.path.exists(build_facade_exe(tmp_path, "mesh_world.cpp"))
    assert os.path.exists(build_facade_exe(tmp_path, "boat_controller.cpp"))


@pytest.mark.gpu
def test_hover_controller_through_body_interface(tmp_path):
    """A HoverCarPhysics-shaped controller drives a body through physics_system->GetBodyInterface() (AddForce, AddTorque,
    GetWorldTransform, GetLinearVelocity ...): the body settles at the spring's target height and has yawed."""
    exe = build_facade_exe(tmp_path, "hover_controller.cpp")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.gpu
def test_boat_controller_on_the_buoyancy_sweep(tmp_path):
    """A BoatPhysics-shaped controller: a box hull floats at its density ratio on think()'s buoyancy sweep, is driven by a thrust applied
    at the propellor point, slowed by drag scaled with PhysicsObject::last_submerged_volume, and turned by a rudder force."""
    exe = build_facade_exe(tmp_path, "boat_controller.cpp")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.gpu
def test_particle_rays_serial_and_batched(tmp_path):
    """A ParticleManager-shaped loop (one traceRay per particle per frame, 2048 particles) against the batched extension traceRays():
    identical results; the printed timings document what the per-call latency costs."""
    exe = build_facade_exe(tmp_path, "particle_rays.cpp")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.gpu
def test_car_controller_through_vehicle_constraint(tmp_path):
    """A CarPhysics-shaped caller builds JPH::VehicleConstraintSettings / WheelSettingsWV / WheeledVehicleControllerSettings as
    CarPhysics.cpp:94-231 does, registers the constraint, drives (throttle, steer right, brake) and reads the wheels back."""
    exe = build_facade_exe(tmp_path, "car_controller.cpp")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.gpu
def test_car_physics_call_sequence(tmp_path):
    """The statements of CarPhysics.cpp:55-231 (constructor), :299-470 (update) and :258-272 (destructor) against the look-alike headers:
    ConvexHullShapeSettings -> OffsetCenterOfMassShapeSettings -> BodyCreationSettings -> BodyInterface::CreateBody / AddBody,
    GetWorldTransform + StoreFloat4x4, the righting torque through Quat::sRotation / Conjugated / GetAxisAngle, GetWheelLocalTransform /
    GetWheelWorldTransform, BodyLockRead, SubShapeID::PopID.  The car drives, is flipped, rights itself."""
    exe = build_facade_exe(tmp_path, "car_physics_sequence.cpp")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "car_physics_sequence: ok" in r.stdout


@pytest.mark.gpu
def test_portal_compound_shape_and_sub_shape_ids(tmp_path):
    """MeshBuilding::makePortalMeshes (MeshBuilding.cpp:377-413: create_tris_for_mat filter + StaticCompoundShapeSettings of the arch mesh
    and the inner box), the portal object of GUIClient.cpp:2379-2393, PlayerPhysics::OnContactAdded (BodyLockRead -> user data) and the
    SubShapeID::PopID test of GUIClient.cpp:6482-6491: walking into the opening touches sub shape 1, walking into a post sub shape 0."""
    exe = build_facade_exe(tmp_path, "portal_walkthrough.cpp")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "portal_walkthrough: ok" in r.stdout


@pytest.mark.gpu
def test_bike_controller_through_motorcycle_controller(tmp_path):
    """A BikePhysics-shaped caller (MotorcycleControllerSettings, raked fork, CastCylinder tester, EnableLeanController): the bike stays
    upright on the straight, leans right in the right-hand turn, shifts up."""
    exe = build_facade_exe(tmp_path, "bike_controller.cpp")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.gpu
def test_player_controller_through_character_virtual(tmp_path):
    """A PlayerPhysics-shaped caller on the JPH::CharacterVirtual look-alike: lands, walks at the commanded speed, takes a 0.3 m step,
    sticks to the floor going down, stops at a wall, jumps, refuses a 66 degree slope, rides a moving platform, pushes a crate."""
    exe = build_facade_exe(tmp_path, "player_controller.cpp")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.gpu
def test_mesh_world_through_the_facade(tmp_path):
    """Height-field terrain + static mesh building built through the facade's shape builders; objects rest on them, rays hit their
    front faces only, the player follows the terrain and stops at the building's wall."""
    exe = build_facade_exe(tmp_path, "mesh_world.cpp")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.gpu
def test_facade_config1_matches_oracle(tmp_path, oracle):
    exe = build_facade_exe(tmp_path)
    descs = scenes.config1_256_boxes()
    scene = tmp_path / "scene.bin"
    descs.tofile(scene)
    out = tmp_path / "out.bin"
    steps = 150
    r = subprocess.run([exe, str(scene), str(steps), str(out), "listener"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    print(r.stdout)
    rec = np.fromfile(out, dtype=np.float32).reshape(-1, 12)
    # oracle: same scene; the facade adds with DontActivate then activateObject() for dynamic bodies
    d2 = descs.copy()
    w = oracle.OracleWorld(max_bodies=65536)
    w.add_batch(d2)
    for _ in range(steps):
        w.step(DT)
    s = w.read_states(0, len(descs))
    live_active = s["active"] != 0
    # activated objects had their transforms read back into PhysicsObject::pos/rot; all bodies via getPosInJolt
    assert np.max(np.abs(rec[:, 7:10] - s["pos"])) <= 1e-4
    assert np.max(np.abs(rec[live_active, 0:3] - s["pos"][live_active])) <= 1e-4
    assert np.max(np.abs(rec[live_active, 3:7] - s["rot"][live_active])) <= 1e-4
    assert np.max(np.abs(rec[:, 10] - s["lin_vel"][:, 0])) <= 1e-3
    head = r.stdout.splitlines()[0].split()
    kv = dict(zip(head[0::2], head[1::2]))
    assert int(kv["objects"]) == 257 and int(kv["newly_activated"]) == 256
    assert int(kv["active"]) == int(live_active.sum())
    assert int(kv["contacts_added"]) > 100 and int(kv["persisted"]) > 1000 and int(kv["ray_hit"]) == 1
    assert "after remove: objects 0" in r.stdout
    w.close()
