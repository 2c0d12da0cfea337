"""The reference's unit of inference, `VGGTQwen3VLM.forward` under no_grad at config C2's batch (B = 6, V = 1, 448 x 448, L = 200), N times -
the run behind `forward_only.forward_mfma_frac` of the bench line, alone, for a kernel trace:

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/fwd -o f -- python3 tools/prof_forward.py 16
    python tools/prof_forward.py --summarize gpurun_out/fwd/*/f_kernel_trace.csv 16 [out.csv]

The summary groups launches by (kernel, grid) and divides by the number of forwards; launches that happen once (weight initialisation,
derived-weight refreshes) are listed with their fractional counts."""
import collections
import csv
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))


def summarize(path, n, out_path=None):
    from prof_window import short
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the forwards are the tail of the trace: find the first launch of the last n repetitions by the patch-embed GEMM's count
    agg = collections.defaultdict(list)
    for r in rows:
        g = (int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
        agg[(short(r["Kernel_Name"]), g)].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    if out_path:                                  # the launch sequence of the last forward (kernel, grid, start offset, duration), in start order
        tail = rows[-(len(rows) // (n + 2)):]
        t0 = int(tail[0]["Start_Timestamp"])
        with open(str(out_path) + ".sequence.txt", "w") as fh:
            for r in tail:
                fh.write("%9.1f %7.1f  %s %sx%s\n" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                                                     short(r["Kernel_Name"]), int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), r["Grid_Size_Y"]))
    out = [("kernel", "grid", "calls_per_forward", "ms_per_forward", "median_us")]
    tot = 0.0
    for k, v in sorted(agg.items(), key=lambda kv: -sum(d for _, d in kv[1])):
        per = len(v) // (n + 1)                     # n timed forwards + 1 warm-up; anything launched fewer than n + 1 times is set-up
        if per == 0:
            continue
        tail = sorted(v)[-per * n:]
        d = sorted(x for _, x in tail)
        ms = sum(d) / 1e3 / n
        tot += ms
        out.append((k[0], "x".join(map(str, k[1])), per, round(ms, 3), round(d[len(d) // 2], 1)))
    out.append(("total kernel time", "", "", round(tot, 3), ""))
    for o in out:
        print(",".join(str(x) for x in o))
    if out_path:
        with open(out_path, "w") as fh:
            csv.writer(fh).writerows(out)


def main():
    if sys.argv[1] == "--summarize":
        return summarize(sys.argv[2], int(sys.argv[3]), sys.argv[4] if len(sys.argv) > 4 else None)
    n = int(sys.argv[1])
    import torch
    import bench
    from vggt_qwen3_amd.qwen3 import Qwen3Config
    from vggt_qwen3_amd.vlm import VGGTQwen3VLM
    dev = torch.device("cuda:0")
    import yaml
    from vggt_qwen3_amd.perceiver import PerceiverConfig
    from vggt_qwen3_amd.vlm import VisionLanguageConfig
    pcfg = PerceiverConfig(**yaml.safe_load((ROOT / "configs" / "perceiver_small.yaml").read_text()))
    model = VGGTQwen3VLM(VisionLanguageConfig(text_model_name="synthetic", vision_ckpt_dir="none", num_vis_tokens=128, geom_tokens=0,
                                              projector_cfg=pcfg, text_config=Qwen3Config.qwen3_4b(), device=str(dev), seed=0))
    model.eval()
    pool = [bench.synthetic_batch(6, 1, 200, 448, 151936, model.image_id, 151643, 198, 1234 + i, dev, False) for i in range(4)]

    def fwd(i):
        b = pool[i % 4]
        with torch.no_grad():
            return model(images=b["pixel_values"], geom_token=b["geom_token"], input_ids=b["input_ids"], attention_mask=b["attention_mask"],
                         labels=b["labels"])
    fwd(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        out = fwd(i + 1)
    torch.cuda.synchronize()
    print("ms per forward %.2f" % ((time.perf_counter() - t0) / n * 1e3))


if __name__ == "__main__":
    main()
