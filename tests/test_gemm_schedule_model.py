"""A small model of the LDS-buffer protocol of the two ping-pong GEMM kernels (pika_amd/csrc/gemm_glds.hip:
gemm_pp and gemm_pp_tn), checked exhaustively over the steady state.  The kernels keep LDS-DMA loads in flight
across barriers and wait with COUNTED s_waitcnt vmcnt(N); whether a fragment read can see a piece that has not
landed (RAW), or a piece can overwrite rows a wave has not finished reading (WAR), is decided by barrier and wait
counts alone -- a clean run on hardware proves nothing about it.  This test states the schedule as data (the same
tables as the comments in the kernels) and verifies, for both wave groups:

  RAW  every read of a region of K-tile k happens in an interval AFTER a barrier that ALL waves pass after the wait
       that retired their pieces of that region (one barrier more for the group that runs one segment behind);
  WAR  every piece written into a buffer region is issued at least two intervals after the last read of that region
       (the read is retired by the lgkmcnt(0) in front of the reader's next MFMA segment, one barrier later);
  CNT  with loads returning in order, `vmcnt(N)` at each wait retires exactly the pieces the schedule relies on.

Timeline: a K-tile is 8 intervals (load segment / MFMA segment x 4 quadrant phases) separated by workgroup barriers;
group 1 runs one interval behind group 0, so segment s of iteration t of group g is absolute interval 8 t + s + g."""
import itertools

L0, M0, L1, M1, L2, M2, L3, M3 = range(8)

# gemm_pp: regions of a K-tile buffer and the load segment (of which group) that reads them
#   A block 2g / 2g+1 = m-half 0 / 1 of wave group g; B is read by every wave in phases 0 and 1
PP_READS = {"A0": [(0, L0)], "A1": [(0, L2)], "A2": [(1, L0)], "A3": [(1, L2)],
            "B": [(0, L0), (0, L1), (1, L0), (1, L1)]}
# fetch cursor: pieces of K-tile k are issued in these (iteration offset from k, load segment) slots, in this order,
# by EVERY wave (B0..B3 are pieces of region B)
PP_ISSUE = [("A0", -2, L2), ("A2", -2, L2), ("B0", -2, L3), ("B1", -2, L3), ("B2", -1, L0), ("B3", -1, L0),
            ("A1", -1, L1), ("A3", -1, L1)]
# waits: (load segment, N) -- placed BEFORE that segment's own issues
PP_WAITS = [(L1, 6), (L3, 4)]

# gemm_pp_tn: reduction rows 0..31 / 32..63 of the A and B images; phases 0, 1 read rows 0..31, phases 2, 3 rows 32..63
TN_READS = {"Alo": [(0, L0), (0, L1), (1, L0), (1, L1)], "Blo": [(0, L0), (1, L0)],
            "Ahi": [(0, L2), (0, L3), (1, L2), (1, L3)], "Bhi": [(0, L2), (1, L2)]}
TN_ISSUE = [("Alo0", -2, L3), ("Alo1", -2, L3), ("Blo0", -1, L0), ("Blo1", -1, L0), ("Ahi0", -1, L1), ("Ahi1", -1, L1),
            ("Bhi0", -1, L2), ("Bhi1", -1, L2)]
TN_WAITS = [(L1, 4), (L3, 4)]


def region(piece):
    return piece.rstrip("0123456789") if piece[0] == "B" or len(piece) > 2 else piece


def check(reads, issue, waits, k=10):
    at = lambda g, t, seg: 8 * t + seg + g                                   # noqa: E731  absolute interval
    # --- the in-order load stream of one wave, iterations 0 .. k+2, with the waits in their places
    stream, wait_events = [], []                                              # (piece, ktile, iteration, seg)
    for t in range(k + 3):
        for seg in (L0, L1, L2, L3):
            for ws, n in waits:
                if ws == seg:
                    wait_events.append((t, seg, n, len(stream)))              # issued so far when the wait executes
            for piece, dt, s in issue:
                if s == seg and t - dt >= 2:                                  # K-tiles 0, 1 come from the prologue
                    stream.append((piece, t - dt, t, seg))
    retired_at = {}                                                           # (piece, ktile) -> (iteration, seg) of the wait
    for t, seg, n, issued in wait_events:
        for piece, kt, _, _ in stream[:max(issued - n, 0)]:
            retired_at.setdefault((piece, kt), (t, seg))
    for reg, rd in reads.items():
        pieces = [p for p, _, _ in issue if region(p) == reg]
        assert pieces, reg
        first_read = min(at(g, k, seg) for g, seg in rd)
        last_read_prev = max(at(g, k - 2, seg) for g, seg in rd)              # same buffer, two K-tiles earlier
        for p in pieces:
            (dt, s), = [(d, sg) for q, d, sg in issue if q == p]
            # CNT + RAW: the wait that retires the piece, then the barrier closing that interval for the LATER group
            assert (p, k) in retired_at, "piece %s of K-tile %d is never waited for" % (p, k)
            wt, wseg = retired_at[(p, k)]
            assert wt <= k, (p, wt)
            all_retired = max(at(g, wt, wseg) for g in (0, 1))                # both groups have passed their wait here
            assert first_read > all_retired, "RAW %s: read in interval %d, retired by all waves in %d" % (p, first_read, all_retired)
            # latency slack of the schedule (informational bound): at least two intervals between issue and wait
            assert at(0, wt, wseg) - at(0, k + dt, s) >= 2, p
            # WAR: the earliest wave issues it at least two intervals after the last read of those rows
            first_issue = min(at(g, k + dt, s) for g in (0, 1))
            assert first_issue >= last_read_prev + 2, "WAR %s: issued in %d, rows last read in %d" % (p, first_issue, last_read_prev)
    # every wait retires something the schedule needs and nothing is waited for that was issued in the same segment
    for t, seg, n, issued in wait_events:
        if t >= 4:
            assert issued - n > 0
            youngest_retired = stream[issued - n - 1]
            assert (youngest_retired[2], youngest_retired[3]) != (t, seg)


def test_gemm_pp_schedule_is_race_free():
    check(PP_READS, PP_ISSUE, PP_WAITS)


def test_gemm_pp_tn_schedule_is_race_free():
    check(TN_READS, TN_ISSUE, TN_WAITS)


def test_the_model_rejects_broken_schedules():
    """The checker is not vacuous: schedules the hardware measurements tempted me with fail it."""
    import pytest
    # (the drained schedule this one replaced -- 3 + 3 + 2 pieces one K-tile ahead, vmcnt(0) in phase 3 -- is slow, not
    # racy: the model accepts it)
    check(PP_READS, [("A0", -1, L0), ("A2", -1, L0), ("B0", -1, L0), ("B1", -1, L1), ("B2", -1, L1), ("B3", -1, L1),
                     ("A1", -1, L2), ("A3", -1, L2)], [(L3, 0)])
    with pytest.raises(AssertionError):   # B refilled in phase 2 of the iteration whose phase 1 (group 1) still reads it
        check(PP_READS, [("A0", -2, L2), ("A2", -2, L2), ("B0", -2, L2), ("B1", -2, L3), ("B2", -1, L0), ("B3", -1, L0),
                         ("A1", -1, L1), ("A3", -1, L1)], [(L1, 6), (L3, 4)])
    with pytest.raises(AssertionError):   # refilling group 1's m-half 1 rows (A3) while it may still be reading them
        check(PP_READS, [("A0", -2, L2), ("A2", -2, L2), ("A1", -2, L3), ("A3", -2, L3), ("B0", -1, L0), ("B1", -1, L0),
                         ("B2", -1, L1), ("B3", -1, L1)], [(L1, 6), (L3, 4)])
    with pytest.raises(AssertionError):   # a count that is one too generous: the late pieces are read unretired
        check(PP_READS, PP_ISSUE, [(L1, 7), (L3, 4)])
    with pytest.raises(AssertionError):   # TN: rows 32..63 refilled in phase 3 of the iteration that still reads them
        check(TN_READS, [("Alo0", -2, L3), ("Alo1", -2, L3), ("Ahi0", -2, L3), ("Ahi1", -2, L3), ("Blo0", -1, L0),
                         ("Blo1", -1, L0), ("Bhi0", -1, L1), ("Bhi1", -1, L1)], [(L1, 4), (L3, 4)])


def test_persistent_tile_walk_covers_every_tile_once():
    """The slot -> tile map of gemm_pp (XCD-contiguous runs, bands of PP_GM tile rows walked column by column) and the
    way persistent workgroups step through the slots (blockIdx.x, + gridDim.x, ...), restated from the kernel:
    every output tile is produced exactly once, for ragged tile grids and any workgroup count the launcher picks."""
    GM = 4

    def tile_of(v, ntiles):
        q, r, xcd, idx = ntiles >> 3, ntiles & 7, v & 7, v >> 3
        return (xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q) + idx

    def tile_mn(tile, nx, ntiles):
        band = tile // (GM * nx)
        inb = tile - band * GM * nx
        rows = min(GM, ntiles // nx - band * GM)
        tn = inb // rows
        return band * GM + inb - tn * rows, tn

    for nx, ny in itertools.product((1, 2, 3, 4, 5, 16, 20, 33), (1, 2, 3, 4, 5, 7, 8, 9, 31, 122, 125)):
        ntiles = nx * ny
        for wgs in (256, 304, 8, 16):
            grid = (wgs & ~7) if (wgs >= 8 and ntiles > wgs) else ntiles        # launch_pp_epi
            seen = []
            for w in range(grid):
                slot = w
                while slot < ntiles:
                    tm, tn = tile_mn(tile_of(slot, ntiles), nx, ntiles)
                    assert 0 <= tm < ny and 0 <= tn < nx
                    seen.append((tm, tn))
                    slot += grid
            assert len(seen) == ntiles and len(set(seen)) == ntiles, (nx, ny, wgs)
            # all tiles of one workgroup stay on its XCD's run (slots share v % 8)
            assert grid % 8 == 0 or grid == ntiles


# ---- the register-staged product loop of the search step (csrc/decode_step.hip, Core<>::run) ------------------------
# A pieces: global -> registers araw[s & 1] (requested TWO steps ahead) -> converted into LDS buffer s & 1 at the end of step
# s - 1; W fragments: requested at the start of step s - 1 into `wnext`, moved into `wreg` at its end.  The tail requests
# clamp to the last step instead of branching.  The model replays the statement order of the loop with tagged buffers.
def _core_loop(steps, a_ahead=2):
    log = []
    araw, lds, wreg, wnext = [None, None], [None, None], None, None
    clamp = lambda s: min(s, steps - 1)
    araw[0] = ("A", 0)
    wreg = ("W", 0)
    lds[0] = araw[0]                                    # stage_a(0, araw[0])
    araw[1] = ("A", clamp(1))
    for st in range(steps):                             # (barrier in front of every step)
        P = st & 1
        if a_ahead == 2:
            assert lds[P ^ 1] is None or lds[P ^ 1][1] <= st, "a request must not clobber registers still to be staged"
            staged_from = araw[P ^ 1]                   # what the end of this step converts: requested during step st - 1
            araw[P] = ("A", clamp(st + 2))
        else:                                           # the one-step-ahead form (same registers requested and staged in a step)
            araw[P ^ 1] = ("A", clamp(st + 1))
            staged_from = araw[P ^ 1]
        wnext = ("W", clamp(st + 1))
        log.append((st, lds[P], wreg))                  # the MFMAs of step st read LDS buffer P and wreg
        lds[P ^ 1] = staged_from                        # buffer P ^ 1: last read in step st - 1, behind a barrier
        wreg = wnext
    return log


def test_search_step_product_loop_feeds_every_step_its_own_operands():
    for steps in range(1, 12):
        for ahead in (1, 2):
            log = _core_loop(steps, ahead)
            assert [st for st, _, _ in log] == list(range(steps))
            for st, a, w in log:
                assert a == ("A", st) and w == ("W", st), (steps, ahead, st, a, w)
