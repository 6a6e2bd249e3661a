"""Where the wave slots of one render_fb launch were idle (reads the dump of a -DU3D_TIMELINE build, see tools/tile_timeline.sh).

Each tile recorded s_memrealtime (100 MHz, device-wide) at entry and exit of its wave, HW_ID / XCC_ID, and the length of its walk.  Per SIMD the
resident-wave count r(t) is swept; with `cap` slots per SIMD (the occupancy pin) the idle slot-time  integral(cap - r)  splits into
  ramp   : before the SIMD is first full,      tail : after it is last full,      middle : in between (refill gaps between consecutive tiles).
Also: the fit  duration = a + b * walk  (fixed cost of a tile against cost per walked entry) and the finish time per XCD.
usage: python tools/tile_timeline.py dump.bin [cap] > summary.json"""
import json
import sys

import numpy as np


def main(path, cap=7):
    a = np.fromfile(path, dtype=np.uint32).reshape(-1, 8)
    a = a[a[:, 1] != 0]
    t0, t1, hw, x = a[:, 0].astype(np.int64), a[:, 1].astype(np.int64), a[:, 2], a[:, 3]
    ta, tb, clk = a[:, 4].astype(np.int64), a[:, 5].astype(np.int64), a[:, 6].astype(np.float64)
    base = t0.min()
    t0, t1, ta, tb = [((v - base) % (1 << 32)) * 0.01 for v in (t0, t1, ta, tb)]   # us
    span = float(t1.max())
    xcc, walk = (x & 15).astype(np.int64), (x >> 4).astype(np.int64)
    simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
    key = ((((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd).astype(np.int64)
    simds = np.unique(key)
    dur = t1 - t0
    A = np.stack([np.ones_like(dur), walk.astype(np.float64)], 1)
    coef, *_ = np.linalg.lstsq(A, dur, rcond=None)
    ramp = tail = mid = busy = 0.0
    peak = []
    first_full, last_full = [], []
    for k in simds:
        m = key == k
        ev = np.concatenate([np.stack([t0[m], np.ones(m.sum())], 1), np.stack([t1[m], -np.ones(m.sum())], 1)])
        ev = ev[np.lexsort((ev[:, 1], ev[:, 0]))]              # at equal times: exits before entries
        r = np.cumsum(ev[:, 1])
        tt = ev[:, 0]
        seg = np.diff(np.append(tt, span))
        pre = tt[0] * cap                                     # nothing resident before the first entry
        full = np.nonzero(r >= cap)[0]
        peak.append(int(r.max()))
        idle = (cap - np.minimum(r, cap)) * seg
        busy += float((np.minimum(r, cap) * seg).sum())
        if len(full) == 0:
            mid += float(idle.sum()) + pre
            continue
        f0, f1 = full[0], full[-1] + 1                         # r < cap again from event f1 on
        first_full.append(float(tt[f0])); last_full.append(float(tt[f1]) if f1 < len(tt) else span)
        ramp += float(idle[:f0].sum()) + pre
        tail += float(idle[f1:].sum())
        mid += float(idle[f0:f1].sum())
    total = cap * span * len(simds)
    per_xcc = {int(c): round(float(t1[xcc == c].max()), 2) for c in np.unique(xcc)}
    # refill gap: on one SIMD, time from an exit to the next entry while the SIMD is below `cap`
    out = {
        "tiles": int(len(t0)), "simds_seen": int(len(simds)), "cus_seen": int(len(np.unique(key >> 2))), "slots_per_simd": cap,
        "peak_resident_per_simd": {"min": int(min(peak)), "max": int(max(peak))},
        "launch_span_us": round(span, 2),
        "slot_time_share": {"busy": round(busy / total, 4), "idle_ramp": round(ramp / total, 4), "idle_middle": round(mid / total, 4), "idle_tail": round(tail / total, 4)},
        "simd_first_full_us": {"median": round(float(np.median(first_full)), 2), "max": round(float(np.max(first_full)), 2)} if first_full else None,
        "simd_last_full_us": {"min": round(float(np.min(last_full)), 2), "median": round(float(np.median(last_full)), 2)} if last_full else None,
        "tile_duration_us": {"mean": round(float(dur.mean()), 2), "p50": round(float(np.median(dur)), 2), "p99": round(float(np.percentile(dur, 99)), 2), "max": round(float(dur.max()), 2)},
        "walk_entries": {"mean": round(float(walk.mean()), 2), "max": int(walk.max())},
        "duration_fit_us": {"fixed_per_tile": round(float(coef[0]), 3), "per_walked_entry": round(float(coef[1]), 4),
                            "fixed_share_of_wave_time": round(float(coef[0] * len(dur) / dur.sum()), 3)},
        "last_exit_per_xcc_us": per_xcc,
        "shader_clock_ghz": {"all": round(float(clk.sum() / dur.sum()) * 1e-3, 3), "first_round": round(float(clk[t0 < 1].sum() / dur[t0 < 1].sum()) * 1e-3, 3),
                             "started_after_20us": round(float(clk[t0 > 20].sum() / dur[t0 > 20].sum()) * 1e-3, 3)},
        "phases_us": {name: {"forward_walk_incl_staging": round(float((ta - t0)[m].mean()), 2), "loss_epilogue": round(float((tb - ta)[m].mean()), 2),
                             "backward_walk_incl_publish": round(float((t1 - tb)[m].mean()), 2), "walk": round(float(walk[m].mean()), 2), "tiles": int(m.sum())}
                      for name, m in (("first_round", t0 < 1), ("started_5_to_100us", (t0 > 5) & (t0 < 100)), ("started_in_the_tail", t0 > np.median(last_full))) if m.any()},
        "entries_per_10us": " ".join(str(int(v)) for v in np.histogram(t0, bins=np.arange(0, span + 10, 10))[0]),
        "exits_per_10us": " ".join(str(int(v)) for v in np.histogram(t1, bins=np.arange(0, span + 10, 10))[0]),
    }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 7)
