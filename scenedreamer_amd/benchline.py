"""The ONE line bench.py prints: a compact, purely numeric extract of the full record.

bench.py measures a great deal (per-tile errors, per-launch tables, gates, the reference's own loop ...).  All of that goes to
`bench_detail.json`; what reaches stdout is `compact(detail)` -- at most LINE_LIMIT bytes of JSON, one line, the last line,
nothing else on stdout (bench.py points file descriptor 1 at stderr for everything but this line).  The driver keeps only a
tail of stdout: a line that does not fit is a headline that was never measured (BENCH_r05: 22.6 KB, `parsed: null`).
No torch import here: tests/test_host_cpu.py builds the line from a canned record on the CPU.
"""
import json
import math
import os

LINE_LIMIT = 6144      # bytes of the printed line, hard (the driver's tail is ~9 KB)
SIG = 5                # significant digits kept for floats


def _num(v, sig=SIG):
    """Floats to `sig` significant digits (ints, bools, None, short strings unchanged); NaN / inf -> None (strict JSON)."""
    if isinstance(v, bool) or v is None or isinstance(v, int):
        return v
    if isinstance(v, float):
        if not math.isfinite(v):
            return None
        if v == 0.0:
            return 0.0
        return float(f"{v:.{sig}g}")
    return v


def _pick(d, keys, rename=None):
    """{k: number} for the keys of `keys` that `d` holds (strings cut to 80 characters)."""
    out = {}
    if not isinstance(d, dict):
        return None
    for k in keys:
        if k in d and d[k] is not None:
            v = d[k]
            if isinstance(v, str):
                v = v[:80]
            elif isinstance(v, (dict, list, tuple)):
                continue
            out[(rename or {}).get(k, k)] = _num(v)
    return out or None


def _get(d, *path):
    for p in path:
        if isinstance(d, dict) and p in d:
            d = d[p]
        elif isinstance(d, (list, tuple)) and isinstance(p, int) and -len(d) <= p < len(d):
            d = d[p]
        else:
            return None
    return d


ROOF_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "frac_finished_samples", "frac_of_sustained", "avg_launch_ms",
             "samples_evaluated", "samples_with_colour_branch", "traffic", "issued_frac_of_peak")
GRID_KEYS = ("bound", "achieved", "peak", "unit", "frac", "avg_launch_ms", "dram_GBps", "effective_GBps", "traffic")
CNN_KEYS = ("bound", "achieved", "peak", "unit", "frac", "avg_ms_in_timed_region", "alone_ms", "frac_alone", "terms3x3")
RVIP_KEYS = ("bound", "achieved", "peak", "unit", "frac", "avg_launch_ms", "traffic")
SKY_KEYS = ("bound", "achieved", "peak", "unit", "frac", "avg_launch_ms")
HEAD_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
             "dtype", "data")
# one number each: where in the full record it lives
SCALARS = {
    "dropin_frames_per_s": ("dropin", "frames_per_s"),
    "config3_frames_per_s": ("other", 3, "frames_per_s"),
    "config5_1gpu_frames_per_s": ("other", 5, "frames_per_s"),
    "config3_max_abs_err_tile": ("other", 3, "max_abs_err_tile"),
    "config5_max_abs_err_tile": ("other", 5, "max_abs_err_tile"),
    "fallback_fp32_frames_per_s": ("fallback_fp32_path_frames_per_s",),
    "floor_frames_per_s": ("floor", "frames_per_s"),
    "floor_max_abs_err": ("floor", "max_abs_err_vs_fp32"),
    "colour_skip_off_frames_per_s": ("colour_skip_off_frames_per_s",),
    "style_setup_ms": ("style_cost", "style_setup_ms"),
    "calibration_ms": ("style_cost", "calibration_ms"),
    "first_frame_ms": ("style_cost", "first_frame_ms"),
    "trajectory40_frames_per_s": ("style_cost", "trajectory40_frames_per_s"),
    "delivered_frames_per_s_uint8_host": ("delivered_frames_per_s_uint8_host",),
    "setup_s": ("setup_s",),
    "broadcast_s": ("broadcast", "broadcast_s"),
    "imbalance": ("config", "bands", "imbalance_max_over_mean"),
}
# dropped first -> last when the line would not fit
DROP_ORDER = ("frame_ms_p10_p50_p90", "stage_ms", "band_ms", "roofline_sky", "roofline_rvip", "setup_s", "delivered_frames_per_s_uint8_host",
              "config3_max_abs_err_tile", "config5_max_abs_err_tile", "roofline_cnn", "roofline_grid_sampler", "detail")


def short_dtype(full):
    """`dtype` is the arithmetic type of the path, not a precision essay: 'f16' MFMA products with f32 accumulation (the
    field / sky MLP as a 3-term split), or 'f32' for the un-fused op sequence."""
    if not isinstance(full, str):
        return full
    if full.startswith("f32 (hash grid) + f16 MFMA"):
        return "f16 (3-term split MFMA, f32 accumulate; f32 hash grid)"
    return full[:64]


def compact(detail, detail_path=None):
    """The dict that is printed.  `detail` is bench.py's full record (what used to be the printed line)."""
    d = dict(detail)
    other = d.get("other_configs")
    d["other"] = {r["baseline_config"]: r for r in other if isinstance(r, dict) and "baseline_config" in r} if isinstance(other, list) else {}
    runs = _get(d, "dropin", "runs")
    if isinstance(runs, list) and runs:       # the reference's default tiling (tile_size 128, frame evaluated once) is the first run
        d["dropin"] = dict(d["dropin"], frames_per_s=runs[0].get("frames_per_s"))
    out = {k: _num(d[k]) for k in HEAD_KEYS if k in d}
    out["dtype"] = short_dtype(d.get("dtype"))
    cfg = d.get("config") or {}
    out["config"] = {k: (cfg[k][:120] if isinstance(cfg[k], str) else cfg[k])
                     for k in ("workload", "baseline_config", "path", "apron", "parallelism", "dist_backend") if cfg.get(k) is not None}
    roof = d.get("roofline")
    if isinstance(roof, dict):
        r = dict(roof)
        if str(r.get("kernel", "")).startswith("field_kernel"):
            r["kernel"] = "field_kernel (placement + hash grid + MLP + compositing, f16 3-term MFMA)"
        r.setdefault("frac_finished_samples", r.get("frac_counting_skipped_colour_branch"))
        out["roofline"] = _pick(r, ROOF_KEYS)
    else:
        out["roofline"] = None
    g = d.get("roofline_grid_sampler")
    if isinstance(g, dict):
        out["roofline_grid_sampler"] = _pick(dict(g, dram_GBps=g.get("dram_GBps_from_profile")), GRID_KEYS)
    c = d.get("roofline_cnn")
    if isinstance(c, dict):
        out["roofline_cnn"] = _pick(dict(c, terms3x3=str(_get(c, "precision_gate", "terms3x3"))), CNN_KEYS)
    out["roofline_rvip"] = _pick(d.get("roofline_rvip"), RVIP_KEYS)
    out["roofline_sky"] = _pick(d.get("roofline_sky"), SKY_KEYS)
    cb = d.get("cpu_baseline")
    if isinstance(cb, dict):
        out["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind"))
        if isinstance(cb.get("sample"), str):
            out["cpu_baseline"]["sample"] = (cb.get("sample_short") or cb["sample"])[:120]
    pr = d.get("precision")
    if isinstance(pr, dict):
        out["precision"] = _pick(pr, ("max_abs_err", "bound"))
        v = _get(pr, "vs_gpu_placement_fp32_path", "max_abs_diff")
        if out["precision"] is not None and isinstance(v, float):
            out["precision"]["vs_gpu_fp32_path"] = _num(v)
    for k, path in SCALARS.items():
        v = _get(d, *path)
        if isinstance(v, (int, float)) and not isinstance(v, bool):
            out[k] = _num(v)
    bm = _get(d, "config", "bands", "band_ms")
    if isinstance(bm, (list, tuple)):
        out["band_ms"] = [_num(float(v), 4) for v in bm]
    if isinstance(d.get("frame_ms_p10_p50_p90"), (list, tuple)):
        out["frame_ms_p10_p50_p90"] = [_num(float(v), 4) for v in d["frame_ms_p10_p50_p90"]]
    if isinstance(d.get("stage_ms"), dict):
        out["stage_ms"] = {k: _num(float(v), 4) for k, v in d["stage_ms"].items()}
    if detail_path:
        out["detail"] = os.path.basename(detail_path)
    out = {k: v for k, v in out.items() if v is not None or k in HEAD_KEYS or k == "roofline"}
    for k in DROP_ORDER:                      # never expected to trigger; a guarantee, not a plan
        if len(dumps(out)) <= LINE_LIMIT:
            break
        out.pop(k, None)
    return out


def dumps(rec):
    return json.dumps(rec, separators=(", ", ": "), allow_nan=False)


def line(detail, detail_path=None):
    """The printed line (no newline).  Raises if even the reduced record does not fit -- never print an unparseable headline."""
    s = dumps(compact(detail, detail_path))
    if len(s.encode()) > LINE_LIMIT or "\n" in s:
        raise ValueError(f"bench line is {len(s.encode())} bytes (limit {LINE_LIMIT})")
    return s


class StdoutGuard:
    """Everything that would reach stdout -- Python prints, C-level prints of libraries, child processes -- goes to stderr
    while the guard is active; `emit` writes to the real stdout.  File-descriptor level (dup2), so the reference loop's
    "Rendering frame ..." prints and any library chatter cannot land between the driver and the JSON line."""

    def __init__(self):
        import sys
        sys.stdout.flush()
        self.real = os.dup(1)
        os.dup2(2, 1)

    def emit(self, text):
        import sys
        sys.stdout.flush()
        sys.stderr.flush()
        os.write(self.real, (text.rstrip("\n") + "\n").encode())

    def close(self):
        import sys
        sys.stdout.flush()
        os.dup2(self.real, 1)
        os.close(self.real)
