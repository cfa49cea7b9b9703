cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -s -k "long_memory or relu_gates or fp8_forward_vs_oracle or rccl or overlapped_exchange or g8 or two_rank or grouped_weight" > gpurun_out/t2.log 2>&1; echo "rc=$?" >> gpurun_out/t2.log
tail -40 gpurun_out/t2.log
python - <<'PY'
import json
for tag in ("g8","g1_gates_mem","g1_gates_nomem","g1_gates_dh50"):
    try:
        d=json.load(open(f"gpurun_out/diag_{tag}.json"))
    except Exception as e:
        print(tag, e); continue
    if tag=="g8":
        print(tag, d["total"], sorted(d["cos"].items(), key=lambda kv: kv[1])[:4])
    else:
        print(tag, d["flipped_gates"], sorted(d["relerr"].items(), key=lambda kv:-kv[1])[:4])
PY
