"""Reporting helpers of bench.py (no GPU, no product code): the compact driver line and the detail record."""
