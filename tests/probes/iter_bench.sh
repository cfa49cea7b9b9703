cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for i in 1 2; do
python bench.py --no-extra --no-decode --no-cpu-baseline --from-iterator | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], json.dumps(d['iterator_fed']))"
done
