#!/bin/bash
# round 4: cache policy x block size of the at-scale GAE launch, as a chain (next_value -> gae -> normalise), hot and cold
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r04; mkdir -p $OUT; rm -f $OUT/gae_policy.jsonl
for N in 1048576 4194304; do
 for B in 128 256; do
  for P in 0 1 2 3 4 5 6 7; do
    CUSRL_GAE_POLICY=$P CUSRL_GAE_BLOCK=$B python $R/scripts/pre_update_chain.py --envs $N --json $OUT/gae_policy.jsonl > /dev/null 2>>$OUT/gae_policy.err
  done
 done
done
python - <<PY
import json
rows=[json.loads(l) for l in open("$OUT/gae_policy.jsonl")]
print("envs     blk pol | hot: nv   gae   norm  chain | cold: nv   gae   norm  chain   (fractions of 8 TB/s; chain in us)")
for r in rows:
    h,c=r["hot"],r["cold"]
    print(f'{r["envs"]:8d} {r["block"]:>3s} {r["policy"]:>3s} | {h["next_value"]["frac"]:.3f} {h["gae"]["frac"]:.3f} {h["normalize"]["frac"]:.3f} {h["chain_us"]:7.1f} | {c["next_value"]["frac"]:.3f} {c["gae"]["frac"]:.3f} {c["normalize"]["frac"]:.3f} {c["chain_us"]:7.1f}')
PY
