#!/usr/bin/env python
"""the detect leg of bench.py alone (decode + NMS, B=128 @ 1280^2)"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
print(json.dumps(bench.detect_leg("cuda:0")))
