"""Builds the native parts of acarsdec_amd in-tree (no JIT cache: the built .so travels to the
GPU box with the repo snapshot).

  lib/libacarsdec_amd.so   HIP kernels (gfx950) + C ABI           -- the product
  lib/acarsdec_gpu         (only where /root/reference exists) the reference's UNCHANGED
                           acarsdec.c/acars.c/output.c/label.c/... linked against compat_msk.c
                           instead of msk.c: the end-to-end drop-in demo
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INC = os.path.join(os.path.dirname(HERE), "include")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libacarsdec_amd.so")
LIB_STAMP = os.path.join(LIBDIR, "libacarsdec_amd_stamp.so")
DEMO = os.path.join(LIBDIR, "acarsdec_gpu")
REF = os.environ.get("ACARSDEC_REF", "/root/reference")
ARCH = "gfx950"


def _run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("build step failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))


def _newer(src, dst):
    return (not os.path.exists(dst)) or any(os.path.getmtime(s) > os.path.getmtime(dst) for s in src)


def hipcc():
    for c in ("/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


LIB_POLY = os.path.join(LIBDIR, "libacarsdec_amd_poly.so")
LIB_LAB = os.path.join(LIBDIR, "libacarsdec_amd_lab.so")

# -ffp-contract=off keeps the reference's separate mul/add roundings.  The -mllvm switches only move
# instructions and shape control flow: the demodulator is one long dependent chain per wave, and
# ILP-first scheduling without machine sinking / branch folding / tail duplication, uniform regions
# left unstructured and small diamonds folded into selects measured 13 % faster per bit
# (1.160 -> 1.005 us; profiles/probe/msk_only.py) than the default heuristics.
MSK_FLAGS = ["-O3", "-ffp-contract=off", "-mllvm", "-amdgpu-sched-strategy=max-ilp", "-mllvm", "-disable-machine-sink",
             "-mllvm", "-disable-branch-fold", "-mllvm", "-disable-tail-duplicate",
             "-mllvm", "-structurizecfg-skip-uniform-regions", "-mllvm", "-phi-node-folding-threshold=4"]
# unit -> (flags, in the product library?).  msk2.hip (the two-wave demodulator: bit-identical and slower) is lab only.
UNITS = [("fir.hip", ["-O3"], True), ("msk.hip", MSK_FLAGS, True), ("msk2.hip", MSK_FLAGS, False), ("synth.hip", ["-O3"], True),
         ("blk.hip", ["-O3"], True), ("acg_api.cpp", ["-O2"], True)]
# which units see which define (the others are compiled once and shared between the libraries)
SEES = {"-DACG_LAB": ("fir.hip", "msk2.hip", "acg_api.cpp"), "-DACG_MSK_STAMP": ("msk.hip", "msk2.hip", "acg_api.cpp"),
        "-DACG_MSK_SINCOS_POLY": ("msk.hip", "msk2.hip")}


def build_lib(force=False, stamp=False, poly=False, lab=False):
    """The library in one of four shapes:
      (default)  lib/libacarsdec_amd.so        THE PRODUCT: the default down-converter kernels (wave-private direct, workgroup-
                                               granular fallback, shared-stream, exact-order, the other sample formats), the
                                               one-wave demodulator, block repair / message split, generators;
      lab=True   lib/libacarsdec_amd_lab.so    + -DACG_LAB: every measurement variant (ACG_FIR_VARIANT 0-4, 6-8, 50-56, 70-73),
                                               the two-wave demodulator (msk2.hip), debug shapes -- loaded by tests and probes only;
      stamp=True lib/libacarsdec_amd_stamp.so  the lab build + -DACG_MSK_STAMP (s_memtime stamps in the demodulator's per-bit loop,
                                               read by profiles/probe/msk_phase_stamps.py);
      poly=True  lib/libacarsdec_amd_poly.so   the lab build with -DACG_MSK_SINCOS_POLY (the mixer's sin/cos as the < 1 ulp Cody-Waite +
                                               fdlibm-kernel evaluation instead of table + rotation); a GPU test runs it beside the
                                               lab build (whose one-wave demodulator is the product's object file) and wants
                                               identical bits and state from both demodulator kernels.
    Only the first is ever loaded by the product path (acarsdec_amd/_capi.py load())."""
    os.makedirs(OBJDIR, exist_ok=True)
    defs = (["-DACG_LAB"] if (lab or stamp or poly) else []) + (["-DACG_MSK_STAMP"] if stamp else []) + (["-DACG_MSK_SINCOS_POLY"] if poly else [])
    out_lib = LIB_STAMP if stamp else LIB_POLY if poly else LIB_LAB if lab else LIB
    hdrs = [os.path.join(CSRC, "acg_internal.h"), os.path.join(CSRC, "msk_common.h"), os.path.join(INC, "acarsdec_amd.h"),
            os.path.abspath(__file__)]   # flags live here
    objs = []
    hc = hipcc()
    for name, flags, in_product in UNITS:
        if not in_product and "-DACG_LAB" not in defs:
            continue
        mine = [d for d in defs if name in SEES[d]]
        tag = "".join("_" + d[6:].lower() for d in mine)          # e.g. fir.hip_lab.o, msk.hip_msk_stamp.o
        src = os.path.join(CSRC, name)
        obj = os.path.join(OBJDIR, name + tag + ".o")
        if force or _newer([src] + hdrs, obj):
            _run([hc, "--offload-arch=" + ARCH, "-std=c++17", "-fPIC", "-I" + INC, "-I" + CSRC] + flags + mine + ["-c", src, "-o", obj])
        objs.append(obj)
    src = os.path.join(CSRC, "host_setup.c")
    obj = os.path.join(OBJDIR, "host_setup.o")
    if force or _newer([src] + hdrs, obj):
        _run(["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-I" + INC, "-c", src, "-o", obj])
    objs.append(obj)
    if force or _newer(objs, out_lib):
        _run([hc, "--offload-arch=" + ARCH, "-shared", "-o", out_lib] + objs + ["-ldl", "-lm"])
    return out_lib


def build_demo(force=False):
    """Reference program, unchanged, on top of the GPU library.  Needs the reference tree."""
    if not os.path.exists(os.path.join(REF, "acars.c")):
        return DEMO if os.path.exists(DEMO) else None
    build_lib()
    ref_units = ["acarsdec.c", "acars.c", "output.c", "label.c", "cJSON.c", "netout.c", "fileout.c"]
    mine = [os.path.join(CSRC, "compat_msk.c"), os.path.join(CSRC, "wav_frontend.c")]
    srcs = [os.path.join(REF, u) for u in ref_units] + mine
    if force or _newer(srcs + [LIB], DEMO):
        _run(["gcc", "-O2", "-w", "-DWITH_SNDFILE", "-I" + REF, "-I" + INC] + srcs +
             ["-o", DEMO, "-L" + LIBDIR, "-lacarsdec_amd", "-Wl,-rpath,$ORIGIN", "-lm", "-lpthread"])
    return DEMO


DEMO_RTL = os.path.join(LIBDIR, "acarsdec_gpu_rtl")


def build_demo_rtl(force=False):
    """Reference acarsdec.c + rtl.c + acars.c + output.c ... UNCHANGED, with compat_msk.c instead of msk.c
    and a file-playing librtlsdr stand-in that hands the buffers to acarsdec_amd_in_callback()."""
    if not os.path.exists(os.path.join(REF, "rtl.c")):
        return DEMO_RTL if os.path.exists(DEMO_RTL) else None
    build_lib()
    demo_dir = os.path.join(CSRC, "demo")
    ref_units = ["acarsdec.c", "acars.c", "rtl.c", "output.c", "label.c", "cJSON.c", "netout.c", "fileout.c"]
    mine = [os.path.join(CSRC, "compat_msk.c"), os.path.join(demo_dir, "demo_rtlsdr_file.c")]
    srcs = [os.path.join(REF, u) for u in ref_units] + mine
    if force or _newer(srcs + [LIB], DEMO_RTL):
        _run(["gcc", "-O2", "-w", "-DWITH_RTL", "-DUSE_AMD_IN_CALLBACK", "-I" + REF, "-I" + INC, "-I" + demo_dir] + srcs +
             ["-o", DEMO_RTL, "-L" + LIBDIR, "-lacarsdec_amd", "-Wl,-rpath,$ORIGIN", "-lm", "-lpthread"])
    return DEMO_RTL


DEMO_SOAPY = os.path.join(LIBDIR, "acarsdec_gpu_soapy")
SOAPY_OLD_TAIL = "\t\tcurrent_index = (current_index + res) % rateMult;\n"


def patched_soapy_source():
    """The reference's soapy.c with the ONE hunk of INTEGRATION.md applied, as text (never written to the repo): the
    per-channel loop of the reader thread (soapy.c:228-254: `int n, i; ... current_index = (current_index + res) % rateMult;`)
    becomes a call of acarsdec_amd_soapy_samples() (compat_msk.c).  Everything else -- initSoapy, chooseFc, the oscillator
    tables, the watchdog, open / close -- is the reference's own text."""
    src = open(os.path.join(REF, "soapy.c")).read()
    a = src.index("\t\tint n, i;\n\t\tint\tlocal_ind;")
    b = src.index(SOAPY_OLD_TAIL, a) + len(SOAPY_OLD_TAIL)
    assert "demodMSK(ch, SOAPYOUTBUFSZ)" in src[a:b] and src.count("demodMSK(") == 1
    return (src[:a] + "\t\t{ extern void acarsdec_amd_soapy_samples(const int16_t *iq, int nsamples);\n"
            "\t\t  acarsdec_amd_soapy_samples(soapyInBuf, res); }          /* acarsdec_amd: replaces soapy.c:228-254 */\n" + src[b:])


def build_demo_soapy(force=False):
    """Reference acarsdec.c + acars.c + output.c ... UNCHANGED and soapy.c with the one-hunk binding, compat_msk.c instead of
    msk.c, and a file-playing SoapySDR stand-in (csrc/demo/demo_soapy_file.c; headers: oracle/stub/SoapySDR).  The patched
    soapy.c goes to the compiler through stdin: no reference text is written anywhere in the repo."""
    if not os.path.exists(os.path.join(REF, "soapy.c")):
        return DEMO_SOAPY if os.path.exists(DEMO_SOAPY) else None
    build_lib()
    demo_dir = os.path.join(CSRC, "demo")
    stub = os.path.join(os.path.dirname(HERE), "oracle", "stub")
    ref_units = ["acarsdec.c", "acars.c", "output.c", "label.c", "cJSON.c", "netout.c", "fileout.c"]
    mine = [os.path.join(CSRC, "compat_msk.c"), os.path.join(demo_dir, "demo_soapy_file.c")]
    srcs = [os.path.join(REF, u) for u in ref_units] + mine
    if force or _newer(srcs + [LIB, os.path.join(REF, "soapy.c"), os.path.abspath(__file__)], DEMO_SOAPY):
        obj = os.path.join(OBJDIR, "soapy_bound.o")
        r = subprocess.run(["gcc", "-O2", "-w", "-DWITH_SOAPY", "-I" + REF, "-I" + INC, "-I" + stub, "-x", "c", "-c", "-", "-o", obj],
                           input=patched_soapy_source(), capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("soapy.c (bound) failed to compile:\n" + r.stderr)
        _run(["gcc", "-O2", "-w", "-DWITH_SOAPY", "-I" + REF, "-I" + INC, "-I" + stub] + srcs + [obj] +
             ["-o", DEMO_SOAPY, "-L" + LIBDIR, "-lacarsdec_amd", "-Wl,-rpath,$ORIGIN", "-lm", "-lpthread"])
        os.remove(obj)
    return DEMO_SOAPY


DEMO_AIR = os.path.join(LIBDIR, "acarsdec_gpu_air")
DEMO_SDRPLAY = os.path.join(LIBDIR, "acarsdec_gpu_sdrplay")


def _build_callback_demo(out, macro, front_end, standin, route_flag, force):
    """Reference acarsdec.c + <front end>.c + acars.c + output.c ... UNCHANGED, compat_msk.c instead of msk.c, and a file-playing
    vendor-library stand-in that hands every transfer to the compat entry point instead of the front end's own callback (the
    one-line binding of INTEGRATION.md without touching the reference source); stub headers: oracle/stub."""
    if not os.path.exists(os.path.join(REF, front_end)):
        return out if os.path.exists(out) else None
    build_lib()
    demo_dir = os.path.join(CSRC, "demo")
    stub = os.path.join(os.path.dirname(HERE), "oracle", "stub")
    ref_units = ["acarsdec.c", "acars.c", front_end, "output.c", "label.c", "cJSON.c", "netout.c", "fileout.c"]
    mine = [os.path.join(CSRC, "compat_msk.c"), os.path.join(demo_dir, standin)]
    srcs = [os.path.join(REF, u) for u in ref_units] + mine
    if force or _newer(srcs + [LIB, os.path.abspath(__file__)], out):
        _run(["gcc", "-O2", "-w", "-D" + macro, "-D" + route_flag, "-I" + REF, "-I" + INC, "-I" + stub] + srcs +
             ["-o", out, "-L" + LIBDIR, "-lacarsdec_amd", "-Wl,-rpath,$ORIGIN", "-lm", "-lpthread"])
    return out


def build_demo_air(force=False):
    return _build_callback_demo(DEMO_AIR, "WITH_AIR", "air.c", "demo_airspy_file.c", "USE_AMD_RX_CALLBACK", force)


def build_demo_sdrplay(force=False):
    return _build_callback_demo(DEMO_SDRPLAY, "WITH_SDRPLAY", "sdrplay.c", "demo_sdrplay_file.c", "USE_AMD_STREAM_CALLBACK", force)


MULTIDEV = os.path.join(LIBDIR, "host_multidev")


def build_multidev(force=False):
    """tests/multidev/host_multidev.c: a plain-C host (gcc, no hipcc) that drives one context per GPU through the C ABI;
    it allocates its device input with the HIP runtime's C API, hence -lamdhip64."""
    build_lib()
    src = os.path.join(os.path.dirname(HERE), "tests", "multidev", "host_multidev.c")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    if force or _newer([src, LIB, os.path.join(INC, "acarsdec_amd.h")], MULTIDEV):
        _run(["gcc", "-O2", "-Wall", "-Wextra", "-I" + INC, "-I" + os.path.join(rocm, "include"), src, "-o", MULTIDEV,
              "-L" + LIBDIR, "-lacarsdec_amd", "-L" + os.path.join(rocm, "lib"), "-lamdhip64",
              "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + os.path.join(rocm, "lib"), "-lm"])
    return MULTIDEV


def build_all(force=False):
    lib = build_lib(force)
    build_lib(force, lab=True)
    build_lib(force, poly=True)
    demo = build_demo(force)
    build_demo_rtl(force)
    build_demo_soapy(force)
    build_demo_air(force)
    build_demo_sdrplay(force)
    build_multidev(force)
    return lib, demo


if __name__ == "__main__":
    print(build_all())
