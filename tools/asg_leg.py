"""the ASG bench leg alone (bench.asg_criterion_ms), for a rocprofv3 kernel trace of its launch sequence"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
print(json.dumps(bench.asg_criterion_ms(torch.device("cuda:0"))))
