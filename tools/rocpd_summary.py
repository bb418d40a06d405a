"""Summarise a rocprofv3 rocpd .db (kernel-trace) as a per-kernel stats table, the same
numbers `rocprofv3 --stats` prints: calls, total/avg/min/max duration (ns), VGPR/LDS/scratch."""
import sqlite3
import sys


def main(path, out=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = cur.execute("""
        select s.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start),
               max(d.end - d.start), max(d.private_segment_size), max(d.group_segment_size),
               max(d.workgroup_size_x), max(d.grid_size_x), max(s.arch_vgpr_count), max(s.accum_vgpr_count),
               max(s.sgpr_count)
        from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
        group by s.kernel_name order by 3 desc""").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["kernel,calls,total_ns,avg_ns,min_ns,max_ns,pct,scratch_B_per_lane,lds_B,wg_size,grid,arch_vgpr,accum_vgpr,sgpr"]
    for r in rows:
        name = r[0].replace(",", ";")
        lines.append("%s,%d,%d,%.0f,%d,%d,%.2f,%s,%s,%s,%s,%s,%s,%s" % (
            name, r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / total, r[6], r[7], r[8], r[9], r[10], r[11], r[12]))
    txt = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(txt)
    sys.stdout.write(txt)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
