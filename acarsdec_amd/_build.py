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
# msk_lean.hip (round 6): the same recipe with machine sinking left on (A/B builds on one box, every switch dropped in turn:
# -0.7 % per bit without -disable-machine-sink, +4 % without max-ilp, +2 % without the structurizer switch, the others nil;
# profiles/r06_msk_lean_builds_ab.txt)
MSK_LEAN_FLAGS = [f for i, f in enumerate(MSK_FLAGS) if not (f == "-disable-machine-sink" or (f == "-mllvm" and MSK_FLAGS[i + 1] == "-disable-machine-sink"))]
# unit -> (flags, in the product library?).  msk2.hip (the two-wave demodulator: bit-identical and slower) is lab only.
UNITS = [("fir.hip", ["-O3"], True), ("fir_mm.hip", ["-O3"], True), ("msk.hip", MSK_FLAGS, True), ("msk_lean.hip", MSK_LEAN_FLAGS, True), ("msk2.hip", MSK_FLAGS, False), ("synth.hip", ["-O3"], True),
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
            os.path.join(INC, "acarsdec_amd_lab.h"), os.path.abspath(__file__)]   # flags live here
    hdrs += sorted(os.path.join(CSRC, "lab", f) for f in os.listdir(os.path.join(CSRC, "lab")))       # the lab-only kernel families fir.hip includes
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
    # The library exports exactly what include/acarsdec_amd.h and include/acarsdec_amd_lab.h declare (a linker version script
    # made from the headers; the kernel launchers, the tuning look-ups and host_setup.c's helpers are local).  The stamp
    # build's two extra entry points are declared in the lab header and simply absent from the other builds.
    vs = os.path.join(OBJDIR, "exports.map")
    vs = os.path.join(OBJDIR, "exports_stamp.map" if stamp else "exports.map")
    names = [n for n in declared_symbols() if stamp or n not in STAMP_ONLY]
    text = "{\n  global:\n" + "".join("    %s;\n" % n for n in names) + "  local: *;\n};\n"
    if not os.path.exists(vs) or open(vs).read() != text:
        with open(vs, "w") as f:
            f.write(text)
    if force or _newer(objs + [vs], out_lib):
        _run([hc, "--offload-arch=" + ARCH, "-shared", "-o", out_lib] + objs + ["-Wl,--version-script=" + vs, "-ldl", "-lm"])
    return out_lib


STAMP_ONLY = ("acg_msk_stamp_read", "acg_msk_lanes_per_channel")


def declared_symbols(headers=("acarsdec_amd.h", "acarsdec_amd_lab.h")):
    """names of the functions the public headers declare, in order (comments stripped; a declaration is `name(` at the start
    of a line after its return type)"""
    import re
    names = []
    for h in headers:
        src = re.sub(r"/\*.*?\*/", "", open(os.path.join(INC, h)).read(), flags=re.S)
        for m in re.finditer(r"^(?:const\s+)?(?:unsigned\s+)?[A-Za-z_][A-Za-z0-9_]*\s*\**\s*(ac(?:g|arsdec_amd)_[A-Za-z0-9_]+)\s*\(", src, flags=re.M):
            if m.group(1) not in names:
                names.append(m.group(1))
    return names


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


DEMO_SOAPY = os.path.join(LIBDIR, "acarsdec_gpu_soapy")
DEMO_AIR = os.path.join(LIBDIR, "acarsdec_gpu_air")
DEMO_SDRPLAY = os.path.join(LIBDIR, "acarsdec_gpu_sdrplay")
COMPAT_INCLUDE = '#include "acarsdec_amd_compat.h"      /* acarsdec_amd */\n'


def _cut(src, start, end, what):
    """(a, b): the span from the first `start` to the end of the first `end` behind it; both must be there exactly as written"""
    assert src.count(start) == 1, "%s: anchor %r occurs %d times" % (what, start, src.count(start))
    a = src.index(start)
    b = src.index(end, a) + len(end)
    return a, b


def patched_source(front_end):
    """The reference's <front_end>.c with the ONE binding hunk of INTEGRATION.md applied, as text (never written to the repo:
    it goes to the compiler through stdin).  Everything else of the file -- init*, chooseFc, the tap / oscillator tables, the
    watchdog, open / close -- is the reference's own text.

      rtl.c      rtl.c:364: librtlsdr gets a callback that keeps the watchdog reset of rtl.c:318-320 and hands the buffer to
                 acarsdec_amd_in_callback() instead of the static in_callback()
      soapy.c    soapy.c:228-254: the per-channel loop of the reader thread becomes acarsdec_amd_soapy_samples(soapyInBuf, res)
      air.c      air.c:293-338: the body of rx_callback() becomes acarsdec_amd_air_samples(samples, sample_count, AIRMULT)
      sdrplay.c  sdrplay.c:213-236: the body of myStreamCallback() becomes acarsdec_amd_sdrplay_samples(xi, xq, numSamples)"""
    src = open(os.path.join(REF, front_end)).read()
    if front_end == "rtl.c":
        old = "\trtlsdr_read_async(dev, in_callback, NULL, 4, rtlInBufSize);\n"
        assert src.count(old) == 1 and src.count("in_callback") == 2          # its definition and this one use
        wrapper = ("static void in_callback_amd(unsigned char *buf, uint32_t nread, void *ctx)      /* acarsdec_amd: replaces in_callback at rtl.c:364 */\n"
                   "{\n\tpthread_mutex_lock(&cbMutex);\n\twatchdogCounter = 50;\n\tpthread_mutex_unlock(&cbMutex);\n"
                   "\tacarsdec_amd_in_callback(buf, nread, ctx);\n}\n\n")
        a = src.index("static void *readThreadEntryPoint(void *arg)")
        src = src[:a] + COMPAT_INCLUDE + wrapper + src[a:]
        return src.replace(old, "\trtlsdr_read_async(dev, in_callback_amd, NULL, 4, rtlInBufSize);\n")
    if front_end == "soapy.c":
        a, b = _cut(src, "\t\tint n, i;\n\t\tint\tlocal_ind;", "\t\tcurrent_index = (current_index + res) % rateMult;\n", front_end)
        assert "demodMSK(ch, SOAPYOUTBUFSZ)" in src[a:b] and src.count("demodMSK(") == 1
        return (src[:a] + "\t\t{ extern void acarsdec_amd_soapy_samples(const int16_t *iq, int nsamples);\n"
                "\t\t  acarsdec_amd_soapy_samples(soapyInBuf, res); }          /* acarsdec_amd: replaces soapy.c:228-254 */\n" + src[b:])
    if front_end == "air.c":
        a, b = _cut(src, "\tfloat* pt_rx_buffer;\n", "        ind=ben;\n", front_end)
        assert src.index("static int rx_callback(airspy_transfer_t* transfer)") < a and "demodMSK(ch,m);" in src[a:b] and src.count("demodMSK(") == 1
        src = (src[:a] + "\tacarsdec_amd_air_samples((const float *)transfer->samples, transfer->sample_count, AIRMULT);"
               "          /* acarsdec_amd: replaces air.c:293-338 */\n" + src[b:])
        a = src.index("int ind=0;\nstatic int rx_callback")
        return src[:a] + COMPAT_INCLUDE + src[a:]
    if front_end == "sdrplay.c":
        a, b = _cut(src, "int n, i;\nint\tlocal_ind;\n", "\tcurrent_index\t= (current_index + numSamples) % SDRPLAY_MULT;\n", front_end)
        assert src.index("void myStreamCallback (") < a and "demodMSK (ch, 512);" in src[a:b] and src.count("demodMSK (") == 1
        src = (src[:a] + "\tacarsdec_amd_sdrplay_samples(xi, xq, (int)numSamples);          /* acarsdec_amd: replaces sdrplay.c:213-236 */\n" + src[b:])
        a = src.index("static\nint current_index = 0;")
        return src[:a] + COMPAT_INCLUDE + src[a:]
    raise ValueError(front_end)


def patched_soapy_source():
    return patched_source("soapy.c")


def _build_bound_demo(out, macro, front_end, standin, force, extra_inc=()):
    """The reference program on one front end, bound to the GPU library the way INTEGRATION.md documents: acarsdec.c + acars.c +
    output.c ... UNCHANGED, <front_end>.c with its one-hunk binding applied (patched_source), compat_msk.c instead of msk.c,
    and a file-playing vendor-library stand-in (csrc/demo/; stub headers: oracle/stub) that knows nothing about the GPU: it
    calls whatever callback / returns whatever read the front end asks for, exactly as for the CPU twin in oracle/_ref."""
    if not os.path.exists(os.path.join(REF, front_end)):
        return out if os.path.exists(out) else None
    build_lib()
    demo_dir = os.path.join(CSRC, "demo")
    stub = os.path.join(os.path.dirname(HERE), "oracle", "stub")
    ref_units = ["acarsdec.c", "acars.c", "output.c", "label.c", "cJSON.c", "netout.c", "fileout.c"]
    mine = [os.path.join(CSRC, "compat_msk.c"), os.path.join(demo_dir, standin)]
    srcs = [os.path.join(REF, u) for u in ref_units] + mine
    incs = ["-I" + REF, "-I" + INC] + ["-I" + i for i in extra_inc] + ["-I" + stub]
    if force or _newer(srcs + [LIB, os.path.join(REF, front_end), os.path.join(INC, "acarsdec_amd_compat.h"), os.path.abspath(__file__)], out):
        obj = os.path.join(OBJDIR, front_end + "_bound.o")
        r = subprocess.run(["gcc", "-O2", "-w", "-D" + macro] + incs + ["-x", "c", "-c", "-", "-o", obj],
                           input=patched_source(front_end), capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("%s (bound) failed to compile:\n%s" % (front_end, r.stderr))
        _run(["gcc", "-O2", "-w", "-D" + macro] + incs + srcs + [obj] +
             ["-o", out, "-L" + LIBDIR, "-lacarsdec_amd", "-Wl,-rpath,$ORIGIN", "-lm", "-lpthread"])
        os.remove(obj)
    return out


def build_demo_rtl(force=False):
    return _build_bound_demo(DEMO_RTL, "WITH_RTL", "rtl.c", "demo_rtlsdr_file.c", force, extra_inc=(os.path.join(CSRC, "demo"),))


def build_demo_soapy(force=False):
    return _build_bound_demo(DEMO_SOAPY, "WITH_SOAPY", "soapy.c", "demo_soapy_file.c", force)


def build_demo_air(force=False):
    return _build_bound_demo(DEMO_AIR, "WITH_AIR", "air.c", "demo_airspy_file.c", force)


def build_demo_sdrplay(force=False):
    return _build_bound_demo(DEMO_SDRPLAY, "WITH_SDRPLAY", "sdrplay.c", "demo_sdrplay_file.c", force)


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
