#!/usr/bin/env python
"""The drop-in's CPU decoder (fpng_decode.cpp: small images, hosts without a GPU, files the GPU path leaves undecided) next to the
reference's decoder (oracle/_ref), one host core, fpng::fpng_decode_memory into a reused vector timed in C++; no GPU needed.
    python tools/cpu_decode_timing.py [swap]      # swap: decode RGB files to RGBA and RGBA files to RGB"""
import sys, os, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,os.path.join(ROOT,'tests')); sys.path.insert(0,ROOT)
os.environ["FPNG_AMD_DECODE_CPU"]="1"
import numpy as np, ctypes as C
import dropin, fpng_amd, ui_images, real_image
from cpu_ref import oracle, ref
R=ref()
photo=real_image.rgb_pixels(R.decode)
cases=[("grad 4K RGBA",np.asarray(fpng_amd.synth_image("grad",3840,2160,4)).reshape(-1),3840,2160,4),
       ("grad 1080p RGB",np.asarray(fpng_amd.synth_image("grad",1920,1080,3)).reshape(-1),1920,1080,3),
       ("photo x4 RGB",np.ascontiguousarray(np.tile(photo,(2,2,1))).reshape(-1),photo.shape[1]*2,photo.shape[0]*2,3),
       ("glyphs 1080p RGB",np.ascontiguousarray(ui_images.glyphs(1920,1080,3)).reshape(-1),1920,1080,3),
       ("512x512 RGB grad",np.asarray(fpng_amd.synth_image("grad",512,512,3)).reshape(-1),512,512,3)]
for name,img,w,h,c in cases:
    for flags in (0,1):
        png=R.encode(img,w,h,c,flags)
        d=(7-c) if len(sys.argv)>1 and sys.argv[1]=='swap' else c
        tr=min(R.time_decode(png,d,5) for _ in range(3))
        tm=min(dropin.time_decode(png,d,reps=5) for _ in range(3))
        print(f"{name:18s} flags {flags}: reference {w*h/tr/1e6:7.1f} MP/s   drop-in CPU tier {w*h/tm/1e6:7.1f} MP/s   ratio {tr/tm:.2f}")
