"""Rank-0 console output (thetis/log.py:43-72 behaviour: only rank 0 prints)."""
import os
import sys


def print_output(msg):
    if int(os.environ.get('RANK', '0')) == 0:
        print(msg)
        sys.stdout.flush()
