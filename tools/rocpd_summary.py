#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd .db (kernel trace) into a CSV like `rocprofv3 --stats` prints: name, calls, total/avg us, %."""
import csv
import sqlite3
import sys


def main(db_path, out_csv):
    db = sqlite3.connect(db_path)
    rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    with open(out_csv, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDuration(us)", "AverageDuration(us)", "Percentage"])
        for r in rows:
            w.writerow([r[0], r[1], f"{r[2]:.3f}", f"{r[3]:.3f}", f"{r[4]:.3f}"])
    print(f"{len(rows)} kernels -> {out_csv}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
