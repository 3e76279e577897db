# digest of the density checksums of the C3 estimators for environment variants (same digest = same bits)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for cfg in "${@}"; do
  echo "== $cfg"
  env $cfg python3 scripts/dev/kde_facts.py 1e7 12 2>&1 | grep -v "^W2026\|^E2026" > gpurun_out/kde_digest.log
  tail -1 gpurun_out/kde_digest.log
  python3 - <<'PY'
import json, hashlib
rows = [json.loads(l) for l in open("gpurun_out/kde_digest.log") if l.startswith('{"c"')]
print("checksums digest", hashlib.md5(",".join(repr(r["checksum"]) for r in rows).encode()).hexdigest(), "create", round(sum(r["create_ms"] for r in rows), 2))
PY
done
