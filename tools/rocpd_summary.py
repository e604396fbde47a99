#!/usr/bin/env python3
"""Dump the per-kernel summary (`top_kernels` view) of a rocprofv3 rocpd SQLite database as CSV.
Usage: tools/rocpd_summary.py gpurun_out/prof_x/x_results.db > profiles/x_kernel_stats.csv"""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [c[1] for c in cur.execute("pragma table_info(top_kernels)")]
w = csv.writer(sys.stdout)
w.writerow(cols + ["unit=us"])
for row in cur.execute("select * from top_kernels"):
    w.writerow(row)
