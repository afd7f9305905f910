"""Per kernel, the mean of every counter of a rocprofv3 --pmc run stored as rocpd SQLite (pmc_results.db): one line per (kernel, counter).

usage: python scripts/pmc_db_summary.py <pmc_results.db> [kernel-name substring]"""
import sqlite3, sys, collections
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
t = lambda key: next(x for x in tabs if x.startswith("rocpd_" + key))
pm, ip, kd, ks = t("pmc_event"), t("info_pmc"), t("kernel_dispatch"), t("info_kernel_symbol")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
q = f"""select s.kernel_name, p.name, count(*), avg(e.value), avg(d.end - d.start)
        from {pm} e join {ip} p on e.pmc_id = p.id join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id
        group by s.kernel_name, p.name order by s.kernel_name, p.name"""
rows = collections.defaultdict(list)
for name, ctr, n, v, dur in c.execute(q):
    if flt in name:
        rows[name].append((ctr, n, v, dur))
for name, lst in rows.items():
    print(f"{name[:150]}   ({lst[0][1]} dispatches, mean {lst[0][3] / 1e3:.1f} us)")
    for ctr, n, v, dur in lst:
        print(f"    {ctr:32s} {v:18.1f}")
