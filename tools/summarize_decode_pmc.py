"""profiles/r4_decode_pmc_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of tools/bench_decode.py: the persistent
decode kernel's bytes per launch against the bytes it has to move (the 36 layers' weights once + the cached K / V rows).
Usage: python tools/summarize_decode_pmc.py FETCH_counter_collection.csv WRITE_counter_collection.csv OUT.json [prompt_len]"""
import csv
import json
import statistics
import sys


def load(path, name):
    return [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if "decode_layers" in r["Kernel_Name"] and r["Counter_Name"] == name]


def main():
    f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    prompt = int(sys.argv[4]) if len(sys.argv) > 4 else 200
    alg = 2 * 36 * (6144 * 2560 + 2560 * 4096 + 19456 * 2560 + 2560 * 9728 + 2 * 2560 + 2 * 128) + 36 * 8 * prompt * 128 * 2 * 2
    out = {"kernel": "decode_layers_kernel<2560, 9728, 32, 8>",
           "command": "rocprofv3 --pmc FETCH_SIZE (and, separately, WRITE_SIZE) -- python3 tools/bench_decode.py --new 16",
           "launches": len(f), "FETCH_SIZE_KiB_median": statistics.median(f), "WRITE_SIZE_KiB_median": statistics.median(w),
           "fetch_bytes_per_launch_corrected": statistics.median(f) * 1024 * 2, "write_bytes_per_launch": statistics.median(w) * 1024,
           "correction": "gfx950: FETCH_SIZE counts 128-byte requests at 64 bytes - doubled (MI355X_MICROARCH.md, HBM / rocprofv3 section); "
                         "WRITE_SIZE as reported",
           "algorithmic_bytes_per_launch": alg,
           "note": "algorithmic = the 36 layers' projection and norm weights once + the cached K / V rows of the prompt",
           "fetch_over_algorithmic": statistics.median(f) * 1024 * 2 / alg}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
