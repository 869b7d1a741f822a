import sys, torch, time
sys.path.insert(0, "/root/repo")
from yolov5m_amd import config
from yolov5m_amd.model import YOLOV5m
from yolov5m_amd.utils.synth import synth_images
m = YOLOV5m(first_out=48, nc=80, anchors=config.ANCHORS, ch=(192, 384, 768)).to("cuda"); m.compute_dtype = "bf16"; m.eval()
x = synth_images(32, 640, 640).to("cuda")
with torch.no_grad():
    for _ in range(12): m(x)
torch.cuda.synchronize()
