"""Runs the two C++ adapter programs on the GPU: the reference's PclOmp convergence test written
against the plain face of include/ndt_hip/ndt_hip.hpp, and the driver-shaped program that goes
through the reference's own names (include/compat) with Eigen / PCL / GTSAM-typed values
(API mocks, tests/cpp/mock)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")


def _run(name):
    exe = os.path.join(CPP, name)
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", CPP, name])
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(p.stdout, p.stderr)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "PASS" in p.stdout


@pytest.mark.gpu
def test_cpp_adapter_reference_convergence():
    _run("test_adapter")


@pytest.mark.gpu
def test_cpp_driver_shaped_calls_through_reference_names():
    """run/pipeline.cpp:464-481,557-604, run/pipeline_ligo_tc.cpp:293,531, include/pipeline.hpp:163-222
    and run/pipeline_lo_svn.cpp:299-320,387-388 as the drivers write them."""
    _run("test_driver_shape")


@pytest.mark.gpu
def test_cpp_grid_queries_and_svn_math_hook_against_the_oracle():
    """extern/svn_ndt/include/voxel_grid_covariance.h:194-200,280-381 and svn_ndt.h:186-206 on the adapter, through the
    compat header names: neighbour sets (DIRECT7 / DIRECT1 / radius) equal the oracle's on 4000 queries, and
    computeParticleDerivatives equals the oracle's derivatives to 1e-9."""
    _run("test_grid_queries")


def test_cpp_adapter_compiles():
    """All faces of the adapter build with plain g++ against the C-ABI library: the dependency-free
    one, and the Eigen + PCL + GTSAM one (against the API mocks) through include/compat."""
    subprocess.check_call(["make", "-C", CPP, "all"])
    assert os.path.exists(os.path.join(CPP, "test_adapter"))
    assert os.path.exists(os.path.join(CPP, "test_driver_shape"))
    assert os.path.exists(os.path.join(CPP, "test_grid_queries"))


def test_compat_headers_cover_the_references_includes():
    """Every pclomp / svn_ndt header include/registercallback.hpp:7-17 names for the NDT path has a
    counterpart under include/compat (the GICP ones are not NDT and stay with the reference)."""
    for rel in ("pclomp/ndt_omp.h", "pclomp/ndt_omp_impl.hpp", "pclomp/voxel_grid_covariance_omp.h",
                "pclomp/voxel_grid_covariance_omp_impl.hpp", "svn_ndt.h", "svn_ndt_impl.hpp",
                "voxel_grid_covariance.h", "voxel_grid_covariance_impl.hpp"):
        assert os.path.exists(os.path.join(ROOT, "include", "compat", rel)), rel
