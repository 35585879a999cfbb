"""print a compact summary of a bench.py JSON line read from stdin"""
import json, sys
d = json.loads(sys.stdin.read())
print(' '.join(sys.argv[1:]), d['value'], d['stage_ms'], 'frac', d['roofline']['frac'])
