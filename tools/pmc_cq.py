"""Summarise the rocprofv3 --pmc passes of the communication-avoiding batched QR (csrc/cq_kernels.h) into
profiles/<round>_pmc_cq.json: bytes per QR against the "read once + written once per level" traffic model, MFMA-pipe
busy fraction and executed MFMA flops per kernel.  Input: gpurun_out/<prefix>_pmc_<shape>_{FETCH_SIZE,WRITE_SIZE,MFMA}/
p_counter_collection.csv written by

    rocprofv3 --pmc FETCH_SIZE  --output-format csv -d ... -o p -- python3 tools/qrbench3.py <shape>      (one pass per counter set)
    rocprofv3 --pmc WRITE_SIZE  ...
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 ...

usage: python tools/pmc_cq.py r04 a 6400x1600x16 16384x4096x1     (round tag, file prefix, shapes)"""
import collections
import csv
import json
import sys

REPS = 3          # tools/qrbench3.py factors every shape three times


def levels(rows):
    out, n = [], (rows + 255) // 256
    while True:
        out.append(n)
        if n == 1:
            return out
        n = (n + 3) // 4


def model_bytes(rows, cols):
    """trailing matrix read once and written once per 64-column block and tree level: level 0 touches every row below the
    block, level l >= 1 the 64-row heads of the level l-1 nodes (256 rows per node of 4 heads)"""
    tot = 0.0
    for jb in range(0, min(rows, cols), 64):
        tcols = max(0, cols - jb - 64)
        nl = levels(rows - jb)
        touched = (rows - jb) + sum(min(256, 64 * nl[l - 1] - 256 * k) for l in range(1, len(nl)) for k in range(nl[l]))
        tot += 2.0 * 8.0 * touched * tcols
    return tot


def main():
    rnd, pre, shapes = sys.argv[1], sys.argv[2], sys.argv[3:]
    out = {"command": "rocprofv3 --pmc <set> --output-format csv -- python3 tools/qrbench3.py <shape>  (separate passes: FETCH_SIZE | "
                      "WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64; 3 factorisations per run)",
           "note": "FETCH_SIZE x 2 (gfx950: 128-byte requests tallied at 64 B, profiles/r03_fetch_write_calibration.txt) + WRITE_SIZE, "
                   "KiB -> bytes, per QR = / (3 repetitions x problems); MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES / 32 x 1024) "
                   "as in profiles/r03_pmc_mfma.json; executed flops = SQ_INSTS_VALU_MFMA_F64 x 2048", "shapes": {}}
    for shape in shapes:
        rows, cols, nprob = (int(v) for v in shape.split("x"))
        agg = collections.defaultdict(float)
        import os
        for c in ("FETCH_SIZE", "WRITE_SIZE", "MFMA", "LDS"):
            f = f"gpurun_out/{pre}_pmc_{shape}_{c}/p_counter_collection.csv"
            if c == "LDS" and not os.path.exists(f):
                continue
            for r in csv.DictReader(open(f)):
                name = r["Kernel_Name"].split("(")[0].replace("void ", "")
                if name.startswith("cq::"):
                    agg[(name, r["Counter_Name"])] += float(r["Counter_Value"])
        kern = sorted(set(k for k, _ in agg))
        per = REPS * nprob
        fetch = sum(agg[(k, "FETCH_SIZE")] for k in kern) * 1024 * 2 / per
        write = sum(agg[(k, "WRITE_SIZE")] for k in kern) * 1024 / per
        insts = sum(agg[(k, "SQ_INSTS_VALU_MFMA_F64")] for k in kern) / per
        algo = 2.0 * rows * cols * cols - 2.0 / 3.0 * cols ** 3
        res = {"matrix_bytes": 8.0 * rows * cols, "fetch_bytes_per_qr": fetch, "write_bytes_per_qr": write,
               "bytes_per_qr": fetch + write, "model_bytes_per_qr": model_bytes(rows, cols),
               "measured_over_model": (fetch + write) / model_bytes(rows, cols),
               "executed_mfma_flops_per_qr": insts * 2048, "householder_flops_per_qr": algo,
               "executed_over_householder": insts * 2048 / algo, "kernels": {}}
        for k in kern:
            busy, sq = agg[(k, "SQ_VALU_MFMA_BUSY_CYCLES")], agg[(k, "SQ_BUSY_CYCLES")]
            res["kernels"][k] = {"fetch_bytes_per_qr": agg[(k, "FETCH_SIZE")] * 2048 / per, "write_bytes_per_qr": agg[(k, "WRITE_SIZE")] * 1024 / per,
                                 "mfma_flops_per_qr": agg[(k, "SQ_INSTS_VALU_MFMA_F64")] * 2048 / per,
                                 "mfma_pipe_busy": busy / (sq / 32.0 * 1024.0) if sq else None,
                                 "lds_bank_conflict_share": (agg[(k, "SQ_LDS_BANK_CONFLICT")] / agg[(k, "SQ_LDS_IDX_ACTIVE")]) if agg.get((k, "SQ_LDS_IDX_ACTIVE")) else None}
        out["shapes"][shape] = res
        print(f"{shape}: {res['bytes_per_qr'] / 1e9:.2f} GB per QR (matrix {res['matrix_bytes'] / 1e6:.0f} MB, model {res['model_bytes_per_qr'] / 1e9:.2f} GB, "
              f"x{res['measured_over_model']:.2f}); executed {res['executed_mfma_flops_per_qr'] / 1e9:.1f} Gflop = {res['executed_over_householder']:.2f} x Householder; "
              + ", ".join(f"{k} busy {v['mfma_pipe_busy']:.2f}" for k, v in res["kernels"].items() if v["mfma_pipe_busy"] is not None))
    json.dump(out, open(f"profiles/{rnd}_pmc_cq.json", "w"), indent=1)


if __name__ == "__main__":
    main()
