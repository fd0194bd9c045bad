"""Timing probe of the NLLB leg alone (gpurun): python scripts/nllb_probe.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
print(json.dumps(bench.translation_leg(0, cpu_check="--cpu" in sys.argv), indent=1))
