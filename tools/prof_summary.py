"""Turn a rocprofv3 results database (--kernel-trace --stats) into a small text summary
that can be committed under profiles/.  Usage: prof_summary.py <results.db> <out.txt> [note]"""
import sqlite3
import sys


def main():
    db_path, out_path = sys.argv[1], sys.argv[2]
    note = sys.argv[3] if len(sys.argv) > 3 else ""
    db = sqlite3.connect(db_path)
    rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    with open(out_path, "w") as f:
        if note:
            f.write(note.rstrip() + "\n\n")
        f.write(f"{'kernel':<64} {'calls':>6} {'total_us':>12} {'avg_us':>12} {'pct':>7}\n")
        for name, calls, total, avg, pct in rows:
            short = name.split("(")[0][-64:]
            f.write(f"{short:<64} {calls:>6} {total:>12.2f} {avg:>12.3f} {pct:>7.2f}\n")
    print(open(out_path).read())


if __name__ == "__main__":
    main()
