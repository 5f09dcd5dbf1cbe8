"""Engine factories for the tests: the real HIP library, and the same sources compiled
against the SIMT emulator (tests/hostsim) for CPU-only checks of the device code."""
import os
import subprocess

from fastp_amd import engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIM_DIR = os.path.join(ROOT, "tests", "hostsim")
SIM_LIB = os.path.join(SIM_DIR, "libfastp_gpu_sim.so")
CSRC = os.path.join(ROOT, "fastp_amd", "csrc")


def _sim_sources():
    out = [os.path.join(SIM_DIR, "sim.cpp"), os.path.join(SIM_DIR, "hip", "hip_runtime.h"),
           os.path.join(ROOT, "include", "fastp_gpu.h")]
    out += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".cpp"))]
    return out


def build_sim():
    if not os.path.exists(SIM_LIB) or any(os.path.getmtime(s) > os.path.getmtime(SIM_LIB) for s in _sim_sources()):
        subprocess.check_call([os.path.join(SIM_DIR, "build.sh")], stdout=subprocess.DEVNULL)
    return SIM_LIB


def sim_engine(params):
    return engine.GpuEngine(params, lib_path=build_sim())


def gpu_engine(params, device=0):
    return engine.GpuEngine(params, device=device)
