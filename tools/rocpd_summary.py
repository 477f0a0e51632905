"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per kernel name x grid: calls, total, avg, % of GPU time.
    python tools/rocpd_summary.py gpurun_out/prof/r1_results.db [> profiles/...txt]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    return name[:90]


def main(path):
    db = sqlite3.connect(path)
    rows = db.execute('select name, grid_x, grid_y, grid_z, workgroup_x, count(*), sum(duration), avg(duration), min(duration), '
                      'max(vgpr_count), max(accum_vgpr_count), max(lds_size) from kernels group by name, grid_x, grid_y, grid_z '
                      'order by sum(duration) desc').fetchall()
    total = sum(r[6] for r in rows)
    span = db.execute('select min(start), max(end) from kernels').fetchone()
    print(f'# rocprofv3 --kernel-trace summary of {path}')
    print(f'# total kernel time {total/1e6:.3f} ms over {sum(r[5] for r in rows)} dispatches; trace span {(span[1]-span[0])/1e6:.3f} ms')
    print(f'{"kernel":92s} {"grid":>18s} {"calls":>7s} {"total_ms":>10s} {"avg_us":>9s} {"min_us":>9s} {"pct":>6s} {"vgpr":>5s} {"agpr":>5s} {"lds":>6s}')
    for name, gx, gy, gz, wx, n, tot, avg, mn, vg, ag, lds in rows:
        grid = f'{gx // max(wx,1)}x{gy}x{gz}'
        print(f'{short(name):92s} {grid:>18s} {n:7d} {tot/1e6:10.3f} {avg/1e3:9.2f} {mn/1e3:9.2f} {100*tot/total:6.2f} {vg or 0:5d} {ag or 0:5d} {lds or 0:6d}')


if __name__ == '__main__':
    main(sys.argv[1])
