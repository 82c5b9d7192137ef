#!/bin/bash
python tools/scratch/placement_probe4.py 1024 8 2>&1 | grep -v "amdgpu.ids"
python tools/scratch/placement_probe3.py 1024 8 2>&1 | grep -v "amdgpu.ids" | head -10
