"""bench.py in parts: the entry point (argument parsing, the case loop, the output) stays in bench.py at the repo root."""
