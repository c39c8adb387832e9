"""The C-side multi-device host (tests/multidev/host_multidev.c) on the GPU box.

One process, one host thread, one acg_ctx per device slot, channel c -> context c mod N (BASELINE configs[3]'s shard, the
loop of rtl.c:344-360 cut N ways).  A 1-GPU box rehearses it with N contexts on device 0: the blocks must be EXACTLY the
single-context blocks, whatever N is and whether the input comes from device or host memory; contexts of equal size must
run at the same rate.  On an 8-GPU node the same binary uses the eight devices (context k on device k mod
acg_device_count()) -- nothing to edit.
"""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
BIN = os.path.join(ROOT, "acarsdec_amd", "lib", "host_multidev")


@pytest.fixture(scope="module")
def job(tmp_path_factory):
    from acarsdec_amd import decoder as D, synth as S, _capi as K
    assert K.load().acg_device_count() > 0 and os.path.exists(BIN), "run __graft_entry__.build() first"
    nch, M, nblk, cb = 50, 200, 8, 2
    rng = np.random.default_rng(20260926)
    rows, taps = [], []
    for c in range(nch):
        a, _ = S.channel_audio(rng, nblk * 1024, gap=(250, 700), text_len=(1, 12), corrupt=[None, "p1", "db", "p1crc"] if c % 3 == 0 else None)   # (some blocks need the repair, some get dropped)
        off = float(rng.integers(-40, 41) * 25000 or 50000)
        rows.append(S.iq_u8_from_envelopes(0.5 * (1 + 0.5 * a)[None, :], M, [off], phases=[rng.uniform(0, 6)], noise=0.02, rng=rng).reshape(-1))
        taps.append(D.rtl_taps(131000000 + int(off), 131000000, M))
    iq = np.stack(rows)
    taps = np.stack(taps).astype(np.float32)
    d = tmp_path_factory.mktemp("multidev")
    iq.tofile(d / "iq.u8")
    taps.tofile(d / "taps.f32")
    return dict(dir=d, nch=nch, M=M, nblk=nblk, cb=cb, iq=iq, taps=taps)


def run(job, N, *extra):
    cmd = [BIN, str(job["dir"] / "iq.u8"), str(job["dir"] / "taps.f32"), str(job["nch"]), str(job["M"]), str(job["nblk"]), str(job["cb"]), str(N)] + list(extra)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-1500:])
    return r.stdout.splitlines(), r.stderr


def test_n_contexts_give_exactly_the_single_context_blocks(job):
    """N in {1, 2, 4} (4 does not divide 50: shards of 13, 13, 12, 12 channels), device-resident and host-fed input: the
    merged (global channel, end_bit)-ordered block list is identical, and equals what one Python-side context decodes."""
    from acarsdec_amd import decoder as D
    one, err1 = run(job, 1)
    assert len(one) >= job["nch"] and "context 0: device 0" in err1
    for N in (2, 4):
        lines, err = run(job, N)
        assert lines == one, N
        assert err.count("context ") >= N and "channels (global" in err
    assert run(job, 3, "--host")[0] == one
    dec = D.Decoder(job["nch"], decim=job["M"], max_blocks=job["nblk"], bitlog=False)
    dec.set_taps(job["taps"])
    dec.in_callback(job["iq"])
    want = ["B %d %d %d %d %02x%02x %s" % (f.chn, f.end_bit, f.len, f.err, f.crc[0], f.crc[1], bytes(f.txt[: max(0, f.len)]).hex())
            for f in dec.drain_frames(16 * job["nch"])]
    dec.close()
    assert one == want


def test_delivered_messages_are_the_single_context_messages(job):
    """--msgs: ACG_F_REPAIR + acg_collect_msgs on every context; N = 4 host-fed == N = 1 device-resident == one Python context,
    and the oracle's block repair + field split agrees on which blocks survive."""
    from acarsdec_amd import decoder as D
    from oracle import oracle as O
    one, _ = run(job, 1, "--msgs")
    assert run(job, 4, "--msgs", "--host")[0] == one and run(job, 2, "--msgs")[0] == one
    dec = D.Decoder(job["nch"], decim=job["M"], max_blocks=job["nblk"], bitlog=False, repair=True)
    dec.set_taps(job["taps"])
    dec.in_callback(job["iq"])
    msgs = dec.drain_msgs(16 * job["nch"])
    want = ["M %d %d %d %s %s %s %s %s" % (m.chn, m.end_bit, m.err, (m.mode or b"-").decode("latin1"), m.addr.decode("latin1"), m.label.decode("latin1"),
                                          (m.bid or b"-").decode("latin1"), bytes(m.txt[: m.txt_len]).hex()) for m in msgs]
    assert one == want and len(one) > job["nch"] // 2
    # the oracle, per channel, on the GPU's own dm
    per = {}
    for m in msgs:
        per.setdefault(int(m.chn), []).append(O.msg_tuple(m))
    for c in range(0, job["nch"], 7):
        ch = O.Channel(c)
        ch.demod(dec.dm(c, job["nblk"] * 1024))
        kept = [b for b in (O.blk_process(f) for f in ch.frames) if b is not None]
        assert per.get(c, []) == [O.msg_tuple(O.msg_split(b)) for b in kept], c
    dec.close()


@pytest.mark.parametrize("N", [2, 4])
def test_contexts_sharing_one_device_all_make_progress_and_their_rates_are_recorded(N):
    """8192 channels of device-generated input over N contexts on the box's GPU(s), timed by the C host: all together, then each
    alone in turn (150 calls = 0.2-0.4 s per context, after a turn for nothing).  The rates go to gpurun_out/multidev_rates.txt
    (committed under profiles/).  What round 4 measured: the spread between equal contexts that share one device (5-18 % with the
    runtime's default of 4 hardware queues per device, whichever context's streams alias being the slow one) is stream -> hardware
    queue aliasing: streams that share a queue serialise.  It does not follow the buffers (--shared-input changes nothing), it does
    follow GPU_MAX_HW_QUEUES (all four together 1.89 M with 4 queues, 2.48 M with 8: profiles/r04_multidev_hw_queues.txt); the host
    sets 16 itself.  A residue of a few per cent remains; on an N-GPU node every device has ONE context.  The test insists that
    every context makes progress at a comparable rate (within 50 %) and that N contexts sharing a device do not collapse."""
    # (--time 150: every context is timed alone over 150 calls after a turn for nothing)
    r = subprocess.run([BIN, "random", "rtl", "8192", "200", "8", "8", str(N), "--msgs", "--time", "150"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-1500:]
    alone = [float(l.split(":")[1].split()[0]) for l in r.stderr.splitlines() if l.startswith("context ") and " alone:" in l]
    together = [float(l.split(":")[1].split()[0]) for l in r.stderr.splitlines() if l.startswith("all ")]
    assert len(alone) == N and len(together) == 1
    spread = max(alone) / min(alone) - 1.0
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "multidev_rates.txt"), "a") as f:
        f.write("N=%d  alone %s  together %s  spread %.2f %%\n" % (N, alone, together, 100 * spread))
    assert spread < 0.50, (alone, spread)          # (observed 5-18 % over eight boxes; this is a sanity bound, the numbers are the result)
    assert together[0] > 0.6 * max(alone)            # N contexts sharing ONE device: no faster than one, not collapsed either
