"""CPU model check of the ping-pong GEMM's LDS-DMA ring schedule (csrc/gemm.hip, gemm_glds_kernel, PP = true).

The K-loop's `s_waitcnt vmcnt(N)` are hand-counted: the LDS-DMA pieces are inline asm hipcc does not track.  Round 4 shipped, for an hour, a
build whose prologue waited for ONE landed stage while the first trip of the new two-stages-per-phase loop read TWO — invisible to 82 GEMM
parity tests (L2-warm operands have always landed), NaN in the training step.  This file restates the CONTROL FLOW of the schedule (prologue,
pair loop, single steady steps, generic steps, the DMA stream running ahead across output tiles) with the worst-case semantics of vmcnt —
after `vmcnt(n)` exactly the n youngest pieces may still be in flight, and nothing lands unless a wait says so — and checks, for every
trip of every tile:
  RAW  every ring stage a trip reads was covered by the wait that ended the PREVIOUS trip's load phase (that is what all eight waves have
       executed before the barrier the reading group passes: the two row groups run one phase apart);
  WAR  a stage is only DMA'd into a slot whose previous occupant was consumed in an EARLIER trip (the lagging group reads one phase late);
  and that the ring never holds more stages than it has slots.
The constants (ring depth, stages in flight, stages landed, the wait table) are parsed out of gemm.hip, so an edit of the source that breaks
the invariant — LAND back to 1, a looser prologue wait — fails here, on the CPU.  The model is checked against itself too: the racy build's
constants must be caught."""
import os
import re

import pytest

SRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cleantransformer_amd", "csrc", "gemm.hip")


def source_constants():
    s = open(SRC).read()
    m = re.search(r"constexpr int FILL = K2 \? NST - (\d+) : NST - (\d+);", s)
    assert m, "FILL definition not found (the model mirrors: constexpr int FILL = K2 ? NST - a : NST - b;)"
    fill_k2, fill_1 = int(m.group(1)), int(m.group(2))
    m = re.search(r"constexpr int LAND = K2 \? (\d+) : (\d+);", s)
    assert m, "LAND definition not found"
    land_k2, land_1 = int(m.group(1)), int(m.group(2))
    m = re.search(r"constexpr int glds_ring\(bool pp, int wm, bool xlane = false\) \{ return !pp \? 3 : \(glds_k2\(pp, wm\) \? (\d+) :", s)
    assert m, "glds_ring definition not found"
    ring_k2 = int(m.group(1))
    # the waits the model assumes, as spelled in the source
    assert "wait_stages(inflight - LAND);" in s, "prologue wait changed: update the model"
    assert "wait_stages(FILL - LAND);" in s, "single steady step wait changed: update the model"
    assert "wait_stages(inflight - 1 - LAND);" in s, "generic step wait changed: update the model"
    assert re.search(r"issue_stage\(wrb\);\s*wrb = [^;]+;\s*issue_stage\(wrb\);\s*wrb = [^;]+;\s*wait_stages\(2\);", s), "pair loop changed: update the model"
    return {"k2": dict(NST=ring_k2, FILL=lambda nst: nst - fill_k2, LAND=land_k2), "k1": dict(FILL=lambda nst: nst - fill_1, LAND=land_1)}


class Race(AssertionError):
    pass


def simulate(tiles, items, NST, FILL, LAND, K2, prologue_land=None):
    """One workgroup's walk.  tiles: K-steps of each output tile it computes, in order; items: K-steps of each DMA work item (the same list
    unless a test wants them to differ).  Returns the number of trips.  Raises Race on a violated invariant."""
    prologue_land = LAND if prologue_land is None else prologue_land

    def clamp(n):                                                        # wait_stages(): the table of spelled-out immediates
        if n >= 4 and NST >= 6:
            return 4
        if n >= 3 and NST >= 5:
            return 3
        return 2 if n >= 2 else (1 if n == 1 else 0)

    issued = 0                                                           # global stage counter: stage ids in issue order
    landed_upto = 0                                                      # stages [0, landed_upto) are known landed (vmcnt is in order)
    consumed_trip = {}                                                   # stage -> trip that read it
    state = {"wi": 0, "ti": 0, "inflight": 0}
    nwork = len(items)
    trip = [0]

    def issue(t):
        nonlocal issued
        s = issued
        old = s - NST
        if old >= 0:
            if old not in consumed_trip:
                raise Race(f"stage {s} overwrites the slot of stage {old}, which was never read (ring overflow)")
            if consumed_trip[old] >= t:
                raise Race(f"WAR: stage {s} issued in trip {t} into the slot of stage {old}, read in trip {consumed_trip[old]} (the lagging group reads one phase later)")
        issued += 1

    def stage_issued():
        state["ti"] += 1
        if state["ti"] == items[state["wi"]]:
            state["wi"] += 1
            state["ti"] = 0

    def wait(n):
        nonlocal landed_upto
        landed_upto = max(landed_upto, issued - clamp(n))

    rd = [0]                                                             # next stage to consume

    def read(k, t):
        for s in range(rd[0], rd[0] + k):
            if s >= issued:
                raise Race(f"trip {t} reads stage {s}, which was never issued")
            if s >= pre_landed[0]:
                raise Race(f"RAW: trip {t} reads stage {s}; the wait before the preceding barrier covered stages < {pre_landed[0]} only")
            consumed_trip[s] = t
        rd[0] += k

    # prologue
    for _ in range(FILL):
        if state["wi"] >= nwork:
            break
        issue(-1)
        stage_issued()
        state["inflight"] += 1
    wait(state["inflight"] - prologue_land)
    pre_landed = [landed_upto]                                           # what the wait that ENDED the previous trip's load phase covered
    for ntc in tiles:
        tc = 0
        while tc < ntc:
            steady = state["inflight"] == FILL and state["wi"] < nwork
            if steady:
                nti = items[state["wi"]]
                if K2:
                    npair = min(ntc - tc, nti - state["ti"]) >> 1
                    if npair > 0:
                        for _ in range(npair):
                            t = trip[0]
                            read(2, t)
                            issue(t)
                            issue(t)
                            wait(2)
                            pre_landed[0] = landed_upto
                            trip[0] += 1
                        state["ti"] += 2 * npair
                        tc += 2 * npair
                        if state["ti"] == nti:
                            state["wi"] += 1
                            state["ti"] = 0
                        continue
                n = min(ntc - tc, nti - state["ti"])
                for _ in range(n):
                    t = trip[0]
                    read(1, t)
                    issue(t)
                    wait(FILL - LAND)
                    pre_landed[0] = landed_upto
                    trip[0] += 1
                state["ti"] += n
                tc += n
                if state["ti"] == nti:
                    state["wi"] += 1
                    state["ti"] = 0
                continue
            # generic step
            t = trip[0]
            read(1, t)
            if state["wi"] < nwork:
                issue(t)
                stage_issued()
                state["inflight"] += 1
            wait(state["inflight"] - 1 - LAND)
            pre_landed[0] = landed_upto
            state["inflight"] -= 1
            tc += 1
            trip[0] += 1
    assert rd[0] == issued == sum(items), (rd[0], issued, sum(items))
    return trip[0]


TILE_WALKS = [[32], [32, 32, 32], [1], [2], [3], [5, 5, 5], [7, 7], [1, 1, 1, 1], [2, 3, 4, 5, 6], [37], [49, 49], [128, 128], [64], [9, 11, 5], [4, 4, 4], [6, 6], [980, 980]]


@pytest.mark.parametrize("walk", TILE_WALKS, ids=lambda w: "x".join(map(str, w)))
def test_two_stages_per_phase_schedule_has_no_raw_or_war_race(walk):
    c = source_constants()["k2"]
    NST = c["NST"]
    trips = simulate(walk, walk, NST, c["FILL"](NST), c["LAND"], K2=True)
    assert trips >= (sum(walk) + 1) // 2


@pytest.mark.parametrize("NST", [4, 5])
@pytest.mark.parametrize("walk", TILE_WALKS, ids=lambda w: "x".join(map(str, w)))
def test_one_stage_per_phase_schedule_has_no_raw_or_war_race(walk, NST):
    c = source_constants()["k1"]
    assert simulate(walk, walk, NST, c["FILL"](NST), c["LAND"], K2=False) == sum(walk)


def test_the_model_catches_the_round4_prologue_race_and_other_broken_schedules():
    c = source_constants()["k2"]
    NST, FILL, LAND = c["NST"], c["FILL"](c["NST"]), c["LAND"]
    assert (NST, FILL, LAND) == (6, 4, 2)
    with pytest.raises(Race, match="RAW"):                               # the first K2 build: prologue waited for one stage, the first pair read two
        simulate([32, 32], [32, 32], NST, FILL, LAND, K2=True, prologue_land=1)
    with pytest.raises(Race, match="RAW"):                               # every wait keeping ONE landed stage (LAND = 1) under the pair loop
        simulate([5, 5, 5], [5, 5, 5], NST, FILL, 1, K2=True)
    with pytest.raises(Race, match="WAR|ring overflow"):                 # five stages in flight on a six-slot ring: the pair overwrites what the lagging group reads
        simulate([32, 32], [32, 32], NST, 5, LAND, K2=True)
    with pytest.raises(Race):                                            # a pair loop on the four-slot ring (two in flight cannot cover the next pair)
        simulate([32], [32], 4, 2, 2, K2=True)
