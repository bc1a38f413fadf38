"""Host-side pieces that need neither a GPU nor ROS: the marker / overlay builders of grid_vision/viz_specs.hpp against
hand-derived known answers (the reference: src/grid_vision_node.cpp:405-523, src/object_detection.cpp:213-224), and the
CMake build of the library (cmake is in the image; hipcc cross-compiles gfx950 without a GPU)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "grid-vision_amd")


def test_marker_and_overlay_specs_known_answers(tmp_path):
    """examples/viz_demo.cpp (plain g++, header only).  Expected values derived by hand from the cited lines:
    one id counter over all markers (:412); a traffic light is a SPHERE (type 2) of 0.3 m living 0.2 s in its colour
    (:421-457); a speed sign is TEXT_VIEW_FACING (type 9), white, scale.z 0.5, one metre above its point (:460-495); an
    UNKNOWN static class gets nothing and burns no id; an L-shape box is a CUBE (type 1) living 0.1 s, colour (0, 0.5, 1),
    pose and scale from the box, scale.z = height = 0 on the PCA path (:499-520).  Overlay: cv::Rect(x_min, y_min,
    x_max - x_min, y_max - y_min) truncates the doubles (100.9 -> 100, 119.8 -> 119, 250.3 -> 250), the label is
    "<class> (<confidence as %f>)", the text sits 5 pixels above the box (:217-222)."""
    exe = str(tmp_path / "viz_demo")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", os.path.join(PKG, "examples", "viz_demo.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    markers = [l for l in lines if l.startswith("marker")]
    want = [
        "marker id 0 ns traffic_light type 2 action 0 life 0.200 frame hero pos 12.500000 -3.250000 4.000000 quat 0.000000000 0.000000000 0.000000000 1.000000000 scale 0.300 0.300 0.300 rgba 1.00 0.00 0.00 1.00 text []",
        "marker id 1 ns traffic_sign type 9 action 0 life 0.200 frame hero pos 20.000000 5.500000 3.250000 quat 0.000000000 0.000000000 0.000000000 1.000000000 scale 0.000 0.000 0.500 rgba 1.00 1.00 1.00 1.00 text [SPEED LIMIT: 60 KMPH]",
        "marker id 2 ns traffic_light type 2 action 0 life 0.200 frame hero pos 7.000000 8.000000 9.000000 quat 0.000000000 0.000000000 0.000000000 1.000000000 scale 0.300 0.300 0.300 rgba 0.00 1.00 0.00 1.00 text []",
        "marker id 3 ns traffic_light type 2 action 0 life 0.200 frame hero pos -1.500000 2.500000 3.500000 quat 0.000000000 0.000000000 0.000000000 1.000000000 scale 0.300 0.300 0.300 rgba 1.00 1.00 0.00 1.00 text []",
        "marker id 4 ns traffic_sign type 9 action 0 life 0.200 frame hero pos 30.000000 -6.000000 2.000000 quat 0.000000000 0.000000000 0.000000000 1.000000000 scale 0.000 0.000 0.500 rgba 1.00 1.00 1.00 1.00 text [SPEED LIMIT: 30 KMPH]",
        "marker id 5 ns lshape_bbox type 1 action 0 life 0.100 frame hero pos 10.000000 2.000000 0.500000 quat 0.000000000 0.000000000 0.382683432 0.923879533 scale 4.500 1.800 1.600 rgba 0.00 0.50 1.00 1.00 text []",
        "marker id 6 ns lshape_bbox type 1 action 0 life 0.100 frame hero pos 25.000000 -4.000000 0.250000 quat 0.000000000 0.100000000 0.000000000 0.990000000 scale 3.200 1.100 0.000 rgba 0.00 0.50 1.00 1.00 text []",
    ]
    assert markers == want
    overlays = [l for l in lines if l.startswith("overlay")]
    assert overlays == [
        "overlay rect 100 50 119 250 text_at 100 45 label [Vehicle (0.950000)] rgb 0 255 0 thickness 2 1 font 0.50",
        "overlay rect 0 3 639 476 text_at 0 -2 label [Person (0.600000)] rgb 0 255 0 thickness 2 1 font 0.50",
        "overlay rect 330 200 130 130 text_at 330 195 label [Unknown (0.123456)] rgb 0 255 0 thickness 2 1 font 0.50",
    ]
    # outline of Rect(100, 50, 119, 250), thickness 2 = offsets -1, 0 around the border lines x = 100 / 218, y = 50 / 299:
    # 4 rows of 120 pixels + 4 columns of 251 pixels - 16 pixels counted twice
    assert lines[-1] == f"drawn green {4 * 120 + 4 * 251 - 16} bbox 99 49 218 299"


@pytest.mark.timeout(900)
def test_cmake_configures_and_builds_the_library(tmp_path):
    """CMakeLists.txt at the repo root (the reference has CMakeLists.txt:9-27 + package.xml): the plain library target,
    no ROS (GV_WITH_ROS2 defaults to OFF), configured and built by cmake with hipcc for gfx950; the library it produces
    exports every function include/gridvision_hip.h declares."""
    if not shutil.which("cmake"):
        pytest.skip("cmake not installed")
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not installed")
    bdir = str(tmp_path / "build")
    cfg = subprocess.run(["cmake", "-S", ROOT, "-B", bdir, "-DCMAKE_BUILD_TYPE=Release"], capture_output=True, text=True, timeout=300)
    assert cfg.returncode == 0, cfg.stdout[-2000:] + cfg.stderr[-2000:]
    bld = subprocess.run(["cmake", "--build", bdir, "--parallel", "8"], capture_output=True, text=True, timeout=800)
    assert bld.returncode == 0, bld.stdout[-2000:] + bld.stderr[-2000:]
    lib = os.path.join(bdir, "libgridvision_hip.so")
    assert os.path.exists(lib)
    nm = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (gv_[a-z0-9_]+)", nm))
    txt = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "gridvision_hip.h")).read(), flags=re.S)
    declared = set(re.findall(r"\b(gv_[a-z0-9_]+)\s*\(", txt))
    assert declared and declared <= exported, sorted(declared - exported)
    assert os.path.exists(os.path.join(bdir, "viz_demo"))   # the host-only example is part of the default build
