#!/bin/bash
# usage (GPU box, repo root): tools/microbench/lds_conflicts.sh <command ...>  -- per kernel: LDS-busy cycles, bank-conflict cycles and their ratio (one --pmc pass)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU --kernel-trace -d $O/pmc_lc -o pmc -- "$@" > $O/pmc_lc.log 2>&1
python - <<'P'
import sqlite3, collections
con = sqlite3.connect("gpurun_out/pmc_lc/pmc_results.db")
agg = collections.defaultdict(dict)
for k, c, v, n in con.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name"):
    name = k.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    agg[name][c] = v / n
for k, d in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_LDS_IDX_ACTIVE", 0)):
    a = d.get("SQ_LDS_IDX_ACTIVE", 0)
    if a < 1e4: continue
    print(f"{k[:44]:44s} lds-busy {a / 1e6:8.2f} M  conflicts {d.get('SQ_LDS_BANK_CONFLICT', 0) / 1e6:8.2f} M ({d.get('SQ_LDS_BANK_CONFLICT', 0) / a:.2f})  lds instr {d.get('SQ_INSTS_LDS', 0) / 1e6:7.2f} M  valu {d.get('SQ_INSTS_VALU', 0) / 1e6:7.2f} M")
P
rm -rf $O/pmc_lc
