#!/usr/bin/env python3
"""Where a generation job's GPU time goes, from a rocprofv3 kernel trace (`--kernel-trace --output-format csv`).

The trace is cut into SEGMENTS at marker kernels -- an image-encoder launch sequence starts at the patch-embedding GEMM's `fill_cls`
neighbour (`fill_cls_kernel`), the visual-token generator at `agg_input_kernel`, a text-tower pass at `text_add_pos_kernel` /
`text_embed_ids_kernel`, cross-validation at the first argmax kernel, inference logits at `scale_f16_kernel` -- and every segment is
reported with its wall span, the time at least one kernel was running (busy), the gaps, and its kernel count.  `--window a,b` (ms from
the first kernel) restricts the report; `--last-step` picks the span from the last `agg_input_kernel`'s enclosing generation to the end.

    python tools/timeline.py <kernel_trace.csv> [--window 120,200] [--out profiles/x.json] [--top 12]
"""
import argparse
import collections
import csv
import json

MARK = (("fill_cls_kernel", "image"), ("agg_input_kernel", "aggregator"), ("text_add_pos_kernel", "text_embedded"),
        ("text_embed_ids_kernel", "text_ids"), ("xval_argmax", "xval"), ("fusion_weights_kernel", "fusion_weights"),
        ("scale_f16_kernel", "inference_logits"))


def kind_of(name):
    for key, kind in MARK:
        if key in name:
            return kind
    return None


def short(name):
    n = name.replace("void (anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "")
    return n[:60]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--window", default="")
    ap.add_argument("--out", default="")
    ap.add_argument("--top", type=int, default=10)
    a = ap.parse_args()
    rows = []
    for r in csv.DictReader(open(a.trace)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                     int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))))
    rows.sort()
    t0 = rows[0][0]
    if a.window:
        lo, hi = (float(x) * 1e6 + t0 for x in a.window.split(","))
        rows = [r for r in rows if lo <= r[0] < hi]
    segs, cur = [], None
    for s, e, name, wg in rows:
        k = kind_of(name)
        # the patch-embedding GEMM precedes fill_cls by one launch: close enough for a span report
        if k is not None and (cur is None or k != cur["kind"] or k in ("image", "text_embedded", "text_ids", "aggregator")):
            if not (cur is not None and k == "xval" and cur["kind"] == "xval"):
                cur = {"kind": k, "start": s, "end": e, "busy": 0, "n": 0, "last_end": s, "kern": collections.Counter()}
                segs.append(cur)
        if cur is None:
            continue
        cur["n"] += 1
        cur["busy"] += max(0, e - max(s, cur["last_end"]))
        cur["last_end"] = max(cur["last_end"], e)
        cur["end"] = max(cur["end"], e)
        cur["kern"][short(name)] += e - s
    out = []
    for i, g in enumerate(segs):
        nxt = segs[i + 1]["start"] if i + 1 < len(segs) else g["end"]
        out.append({"kind": g["kind"], "t_ms": round((g["start"] - t0) / 1e6, 3), "span_us": round((nxt - g["start"]) / 1e3, 1),
                    "busy_us": round(g["busy"] / 1e3, 1), "kernels": g["n"],
                    "top": [(k, round(v / 1e3, 1)) for k, v in g["kern"].most_common(3)]})
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0])
    for o in out:
        x = agg[o["kind"]]
        x[0] += 1; x[1] += o["span_us"]; x[2] += o["busy_us"]; x[3] += o["kernels"]
    print(f"{'kind':18s} {'segments':>8s} {'span ms':>10s} {'busy ms':>10s} {'kernels':>8s}")
    for k, x in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:18s} {x[0]:8d} {x[1] / 1e3:10.3f} {x[2] / 1e3:10.3f} {x[3]:8d}")
    for o in out[-a.top * 4:] if a.window == "" else out:
        print(f"{o['t_ms']:10.3f} ms  {o['kind']:16s} span {o['span_us']:9.1f} us  busy {o['busy_us']:9.1f}  n {o['kernels']:4d}  {o['top']}")
    if a.out:
        json.dump({"segments": out, "by_kind": {k: {"segments": x[0], "span_ms": round(x[1] / 1e3, 3), "busy_ms": round(x[2] / 1e3, 3),
                                                    "kernels": x[3]} for k, x in agg.items()}}, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
