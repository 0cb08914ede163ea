"""In-tree build of libcsp_minsnap.so (HIP kernels + C-ABI) for gfx950.

    python cs-pathplan_amd/build.py [--force]

hipcc cross-compiles without a GPU.  The shared object stays next to this file so it travels
with the repo snapshot to the GPU box (it is git-ignored, not gpurun-ignored).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcsp_minsnap.so")
SOURCES = ["minsnap_capi.hip", "minsnap_generic.hip", "minsnap_chunked.hip", "minsnap_mixed.hip", "minsnap_twist.hip", "minsnap_twist_f32.hip", "minsnap_twist_f32s.hip", "minsnap_twist_f64.hip", "minsnap_twist_f64s.hip", "minsnap_span.hip", "minsnap_fixed.hip", "minsnap_fixed_o2.hip", "minsnap_fixed_o3.hip",
           "minsnap_fixed_o4a.hip", "minsnap_fixed_o4b.hip", "minsnap_fixed_o5.hip",
           "minsnap_fixedpath_o2.hip", "minsnap_fixedpath_o3.hip", "minsnap_fixedpath_o4a.hip", "minsnap_fixedpath_o4b.hip", "minsnap_timealloc.hip", "minsnap_plan.hip", "geo.hip", "alt.hip", "bezier.hip"]
HEADERS = ["minsnap_device.h", "minsnap_launch.h", "minsnap_hoststage.h", "minsnap_timealloc.h", "minsnap_tables.h", "minsnap_fixed_impl.h", "minsnap_fixed_path_impl.h", "minsnap_iface.h", "minsnap_chunked_impl.h", "minsnap_mixed.h", "minsnap_twist_impl.h", "minsnap_twist_launch.h", "minsnap_shard_schedule.h",
           os.path.join("..", "..", "include", "csp_minsnap.h"), os.path.join("..", "..", "include", "csp_geo.h"), os.path.join("..", "..", "include", "csp_alt.h"), os.path.join("..", "..", "include", "csp_bezier.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-Wall", "-Wno-unused-function"]
OBJDIR = os.path.join(HERE, "build")


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def _compile_all(extra, tag, verbose):
    """One hipcc -c per translation unit, in parallel (the per-order kernel files dominate), then link."""
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(OBJDIR, exist_ok=True)
    jobs = []
    for src in SOURCES:
        obj = os.path.join(OBJDIR, "%s%s.o" % (os.path.splitext(src)[0], tag))
        srcp = os.path.join(CSRC, src)
        deps = [srcp] + [os.path.join(CSRC, h) for h in HEADERS]
        if os.path.exists(obj) and all(os.path.getmtime(d) <= os.path.getmtime(obj) for d in deps):
            jobs.append((None, obj))
        else:
            jobs.append(([HIPCC] + FLAGS + extra + ["-c", srcp, "-o", obj], obj))

    def run(job):
        cmd, obj = job
        if cmd:
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        return obj
    with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as ex:
        return list(ex.map(run, jobs))


def build(force=False, verbose=False, stamps=False):
    """stamps=True builds the DIAGNOSTIC library libcsp_minsnap_stamps.so (in-kernel s_memtime
    stamps, never timed, never shipped) next to the product library."""
    if stamps:
        out = os.path.join(HERE, "libcsp_minsnap_stamps.so")
        objs = _compile_all(["-DCSP_STAMPS"], "_stamps", verbose)
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + ["-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"])
        return out
    if force and os.path.isdir(OBJDIR):
        for f in os.listdir(OBJDIR):
            if f.endswith(".o") and not f.endswith("_stamps.o"):
                os.remove(os.path.join(OBJDIR, f))
    if not (force or _stale()):
        return LIB
    objs = _compile_all([], "", verbose)
    # RCCL: csp_minsnap_solve_batch_sharded with device memory scatters / gathers over xGMI (minsnap_capi.hip, RcclTransport)
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"])
    return LIB


def build_host_check(out=None):
    """Compiles the C++ class shim (host/math_util/{minimum_snap,bezier}.hpp) against the bundled
    mini matrix type and links it with the C-ABI library: the 'does the drop-in compile' check."""
    out = out or os.path.join(HERE, "host", "shim_selftest")
    src = os.path.join(HERE, "host", "shim_selftest.cpp")
    cmd = ["g++", "-std=c++14", "-O1", "-Wall", "-I", os.path.join(HERE, "..", "include"),
           "-I", os.path.join(HERE, "host"), src, "-o", out, "-L", HERE, "-lcsp_minsnap",
           "-Wl,-rpath," + HERE, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64"]
    subprocess.check_call(cmd)
    return out


def build_tsan_check(out=None):
    """ThreadSanitizer build of the C-ABI's host code (CPU only): minsnap_capi.hip compiled host-only with
    -fsanitize=thread, linked with the ordinary kernel objects and host/tsan_driver.cpp.  Returns the executable;
    tests/test_sanitizers.py runs it (exit code 66 = a data race was reported)."""
    build()
    out = out or os.path.join(OBJDIR, "tsan_driver")
    tsan_obj = os.path.join(OBJDIR, "minsnap_capi_tsan.o")
    common = ["-O1", "-g", "-std=c++17", "-fPIC", "-fsanitize=thread", "-Wno-option-ignored"]
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "--offload-host-only"] + common +
                          ["-c", os.path.join(CSRC, "minsnap_capi.hip"), "-o", tsan_obj])
    others = [os.path.join(OBJDIR, os.path.splitext(s)[0] + ".o") for s in SOURCES if s != "minsnap_capi.hip"]
    subprocess.check_call([HIPCC, "--offload-arch=gfx950"] + common +
                          ["-x", "c++", os.path.join(HERE, "host", "tsan_driver.cpp"), "-x", "none", tsan_obj] + others +
                          ["-I", os.path.join(HERE, "..", "include"), "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           "-o", out, "-L/opt/rocm/lib", "-lamdhip64", "-lrccl", "-Wl,-rpath,/opt/rocm/lib", "-pthread"])
    return out


if __name__ == "__main__":
    if "--stamps" in sys.argv:
        print(build(stamps=True))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
