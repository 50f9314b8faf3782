#!/bin/bash
# Round 6: address-unit / L1 counters of the 16-bit 64 MiB launches side by side (channel-last `lastaxis_kernel`, `shortrows_kernel` on the same
# bytes, the 52 MB channel-last shape): how busy the texture-address path and the L1 are per dispatch.  One --pmc pass per group, bench.py as the
# program, nothing but --kernel-trace beside --pmc.  -> gpurun_out/r06/ta/<name>_<group>.csv, summary on stdout.
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06/ta; mkdir -p $O; cd /tmp
pass() {   # name, group index, counters, bench args...
  local name=$1 i=$2 grp=$3; shift 3
  rm -rf /tmp/ta_${name}_$i
  timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/ta_${name}_$i -- python3 $R/bench.py --no-cpu --no-batched-extra --no-sharded-extra --evidence-launches 0 --prewarm-seconds 0.2 --steps 60 --warmup 10 "$@" > $O/${name}_$i.log 2>&1
  local f=$(find /tmp/ta_${name}_$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then head -1 $f > $O/${name}_$i.csv; grep -E "mctq" $f | tail -400 >> $O/${name}_$i.csv; return 0; fi
  return 1
}
run() {    # name, bench args...
  local name=$1; shift
  local i=0
  for grp in "GRBM_TA_BUSY GRBM_GUI_ACTIVE SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
             "TCP_GATE_EN1 TCP_TOTAL_CACHE_ACCESSES TCP_TCP_TA_DATA_STALL_CYCLES TCP_READ_TAGCONFLICT_STALL_CYCLES" \
             "TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_TCP_LATENCY TCP_TOTAL_READ" \
             "TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_REQUEST TCP_TA_TCP_STATE_READ"; do
    i=$((i+1))
    sum=$(echo $grp | sed -E 's/\b(TCP_[A-Z_0-9]+)\b/\1_sum/g')
    pass $name $i "$sum" "$@" || pass $name $i "$grp" "$@" || { echo "$name group $i: no counter file"; tail -3 $O/${name}_$i.log; }
  done
}
run lastaxis_4096x4096_axis1_bf16 --config sym --shape 4096x4096 --axis 1 --dtype bf16
run shortrows_4096x4096_axis0_bf16 --dtype bf16
run lastaxis_65536x200_axis1_bf16 --config sym --shape 65536x200 --axis 1 --dtype bf16
run shortrows_65536x200_axis0_bf16 --config sym --shape 65536x200 --axis 0 --dtype bf16
python3 - <<PY
import csv, glob, os, collections
for f in sorted(glob.glob("$O/*.csv")):
    acc = collections.defaultdict(list)
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        k = r.get("Kernel_Name", "")
        if "lastaxis_kernel" in k or "shortrows_kernel" in k:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(os.path.basename(f), {k: round(sum(v) / len(v), 1) for k, v in acc.items()}, "dispatches", max((len(v) for v in acc.values()), default=0))
PY
