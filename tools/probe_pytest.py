"""Diagnostic (run with pytest explicitly): race_probe2's body as a pytest test."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.argv = [sys.argv[0], "0"]
import race_probe2 as R

def test_body():
    R.body(3, 1, 30, "[3-1]")
    R.body(5, 3, 80, "[5-3]")
