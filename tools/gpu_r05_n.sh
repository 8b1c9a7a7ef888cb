#!/bin/bash
# round 5, GPU call N: partial pieces stored with one 16-byte write-through store, and the double-buffered generic decode kernel -- each against the
# PREVIOUS binary (tools/probes/bisect/libatoma_hip_prev.so), interleaved; WRITE_SIZE of the ragged launch; parity of everything that stores pieces
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r05n; mkdir -p $O
PREV=tools/probes/bisect/libatoma_hip_prev.so; NEW=atoma-infer_amd/lib/libatoma_hip.so
echo "== decode shapes, previous vs new"; for lib in $PREV $NEW $PREV $NEW; do n=$([ $lib = $PREV ] && echo prev || echo new); for shape in "C2c decode ragged" "ragged U[2048,4096] MHA" "B=64 h=8" "B=16 S=8192" "B=1 S" "C2a decode" "d=96" "d=256 B=256"; do ATOMA_HIP_LIB=$lib ATOMA_BENCH_DECODE_SHAPE="$shape" timeout 150 python tools/bench_kernels.py decode 2>&1 | grep workload | cut -c1-190 | sed "s/^/[$n] /"; done; ATOMA_HIP_LIB=$lib timeout 300 python tools/rank_step.py 2>&1 | tail -1 | cut -c1-200 | sed "s/^/[$n] /"; done | tee $O/partial_store16_ab.txt
echo "== WRITE_SIZE of the ragged launch and of the B=64 shard, new binary"; cd /tmp; for shape in "C2c decode ragged" "B=64 h=8"; do ATOMA_BENCH_DECODE_SHAPE="$shape" timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w -o c -- python $GRAFT_REPO_ROOT/tools/bench_kernels.py decode > /dev/null 2>&1; python - "$shape" <<'PY'
import csv, glob, sys, os, collections
agg = collections.defaultdict(list)
for f in glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r05n/w/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "paged_decode" in r["Kernel_Name"]: agg[r["Kernel_Name"].split("(")[0][:70]].append(float(r["Counter_Value"]))
for k, v in agg.items(): print(sys.argv[1], "|", k, "WRITE_SIZE KiB per launch", round(sum(v) / len(v), 1))
PY
rm -rf $O/w; done | tee $O/write_size_new.txt; cd $GRAFT_REPO_ROOT
echo "== parity"; timeout 900 python -m pytest tests/test_decode_gpu.py tests/test_kv_fp8_gpu.py tests/test_sync_ticket_gpu.py tests/test_graph_capture_gpu.py tests/test_decode_step_gpu.py tests/test_attention_golden_gpu.py -q -m gpu -x 2>&1 | tail -5 | tee $O/parity.txt
