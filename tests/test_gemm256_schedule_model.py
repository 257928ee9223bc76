"""Happens-before model of the ping-pong schedule of gemm256_kernel<.., SCHED = 2> (mmf_amd/csrc/gemm.hip): the per-wave
event sequence of the kernel is written down once more here, and the two LDS hazards of a three-buffer LDS-DMA ring are
checked for every pair of waves and any number of K-tiles:

  RAW  a wave may read stage t only after EVERY wave has retired its own stage-t LDS-DMA pieces (counted s_waitcnt vmcnt —
       the counter retires in issue order) and a workgroup barrier the reader has passed came after that;
  WAR  a wave may issue LDS-DMA into the buffer of stage t - 3 only after EVERY wave has retired its last reads of that
       stage (s_waitcnt lgkmcnt(0)) and a barrier came after that.

Waves synchronise only through `s_barrier`: the k-th barrier a wave executes is the k-th barrier of every other wave.  This is
a model of the SOURCE (keep it in step with the kernel), not a hardware test; the hardware race screen is
tests/test_gemm256_gpu.py with debug_flags bit 16."""
import pytest


def wave_events(group, nk):
    """[(kind, payload)] in program order.  kinds: issue(stage, pieces) / wait_vm(keep) / read(stage) / wait_lgkm / barrier."""
    ev = [("issue", (0, 6))]
    if nk > 1:
        ev.append(("issue", (1, 6)))
    ev.append(("wait_vm", 6 if nk > 1 else 0))
    ev.append(("barrier", None))
    if group == 1:
        ev.append(("barrier", None))              # the second group runs one interval behind
    for kt in range(nk):
        more = kt + 2 < nk
        for kk in (0, 1):
            ev.append(("read", kt))                                  # LOAD kk: 8 fragment reads of stage kt
            if more:
                ev.append(("issue", (kt + 2, 3)))                    # its half of the six pieces of stage kt + 2
            ev.append(("wait_lgkm", None))
            if kk == 1:
                ev.append(("wait_vm", 6 if more else 0))
            ev.append(("barrier", None))
            ev.append(("barrier", None))                             # COMP kk lies between these two barriers
    if group == 0:
        ev.append(("barrier", None))
    ev.append(("barrier", None))                                     # __syncthreads before the C stage reuses the ring
    return ev


def analyse(group, nk):
    """Per wave: barriers executed before each read / issue, and the barrier count at which each stage's pieces / reads retire."""
    queue, bar = [], 0
    reads, issues, dma_retired, read_retired = [], [], {}, {}
    pending_reads = set()
    for kind, x in wave_events(group, nk):
        if kind == "barrier":
            bar += 1
        elif kind == "issue":
            stage, n = x
            queue += [stage] * n
            issues.append((stage, bar))
        elif kind == "wait_vm":
            done, queue = (queue[:len(queue) - x], queue[len(queue) - x:]) if x else (queue, [])
            for s in set(done):
                if s not in queue:
                    dma_retired.setdefault(s, bar)
        elif kind == "read":
            assert x in dma_retired, "group %d reads stage %d before retiring its own pieces" % (group, x)
            reads.append((x, bar))
            pending_reads.add(x)
        elif kind == "wait_lgkm":
            for s in pending_reads:
                read_retired[s] = bar                                  # last retirement wins
            pending_reads = set()
    assert not queue
    for s in pending_reads:          # reads never explicitly waited for retire, at the latest, when the wave ends
        read_retired[s] = bar
    return dict(bar=bar, reads=reads, issues=issues, dma_retired=dma_retired, read_retired=read_retired)


@pytest.mark.parametrize("nk", [1, 2, 3, 4, 5, 12, 48])
def test_ping_pong_schedule_has_no_lds_hazard(nk):
    w = [analyse(g, nk) for g in (0, 1)]
    assert w[0]["bar"] == w[1]["bar"], "the two groups must execute the same number of barriers"
    for x in (0, 1):
        for y in (0, 1):
            for stage, c_x in w[x]["reads"]:                           # RAW
                assert w[y]["dma_retired"][stage] + 1 <= c_x, ("RAW", nk, x, y, stage)
            for stage, c_x in w[x]["issues"]:                          # WAR against the stage that used this buffer before
                prev = stage - 3
                if prev >= 0:
                    assert w[y]["read_retired"][prev] + 1 <= c_x, ("WAR", nk, x, y, stage)
    # every stage is staged once per wave group member and read in both halves
    for g in (0, 1):
        assert sorted({s for s, _ in w[g]["reads"]}) == list(range(nk))
        assert sorted({s for s, _ in w[g]["issues"]}) == list(range(nk))


def test_the_model_catches_a_broken_schedule(monkeypatch):
    """Sanity of the checker itself: dropping the lgkmcnt(0) that ends a LOAD phase must show up as a WAR hazard."""
    import tests.test_gemm256_schedule_model as M
    orig = M.wave_events

    def broken(group, nk):
        ev, out, seen = orig(group, nk), [], 0
        for e in ev:
            if e[0] == "wait_lgkm":
                seen += 1
                if seen % 2 == 0:       # keep the wait of LOAD0, drop the one of LOAD1
                    continue
            out.append(e)
        return out

    monkeypatch.setattr(M, "wave_events", broken)
    with pytest.raises(AssertionError, match="WAR"):
        M.test_ping_pong_schedule_has_no_lds_hazard(6)
