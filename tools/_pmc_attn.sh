set -e
cd /tmp && export TMPDIR=/tmp
for v in e_one_wg f_three_wg; do
  SRK_LIB_PATH=/root/repo/tpu_superresolution_amd/_variants/$v.so timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d /root/repo/gpurun_out/pmcA_$v -- python3 /root/repo/tools/infer_fwd.py > /dev/null 2>/root/repo/gpurun_out/pmcA_$v.err
  SRK_LIB_PATH=/root/repo/tpu_superresolution_amd/_variants/$v.so timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM --output-format csv -d /root/repo/gpurun_out/pmcB_$v -- python3 /root/repo/tools/infer_fwd.py > /dev/null 2>/root/repo/gpurun_out/pmcB_$v.err
done
cd /root/repo
python - <<'PY'
import csv, glob, collections
for v in ("e_one_wg", "f_three_wg"):
    for tag in ("pmcA", "pmcB"):
        fs = glob.glob(f"gpurun_out/{tag}_{v}/*/*counter_collection.csv")
        if not fs:
            print(v, tag, "no file"); continue
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(fs[0])):
            k = r["Kernel_Name"]
            if "qkv_attn" not in k and "mlp_fused" not in k: continue
            k = k.split("(")[0][-28:]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        for k, d in acc.items():
            print(v, tag, k, {c: f"{x:.3e}" for c, x in d.items()})
PY
