import os, sys, ctypes as C, numpy as np
sys.path.insert(0, os.getcwd())
import torch, torch.distributed as dist
import pcrhpg24_amd as P
from pcrhpg24_amd import dist as pdist
from tests import oracle, scenes
from tests.test_dist_native import dist_lib
W,H=640,360
image,_ = scenes.synth_stream(2_000_000)
of = oracle.OracleFile(image.view())
if "native" in sys.argv:
    lib = dist_lib()
    p = scenes.with_flags(scenes.cameras(W, H)["overview"], lod_percent=100, cull=0)
    r = P.Renderer(W, H, device=0)
    d = C.c_void_p()
    P.HuffmanLasData.create(image).load_all(r)
    ident = C.create_string_buffer(128)
    assert lib.pcr_dist_unique_id(ident) == 0
    assert lib.pcr_dist_create(r.ctx.h, ident, 0, 1, C.byref(d)) == 0
    assert lib.pcr_dist_frame_basic(d, C.byref(p), 0) == 0
    r.ctx.synchronize()
    lib.pcr_dist_destroy(d); r.ctx.close()
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29541")
dev = torch.device("cuda",0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
ctx = P.Context(0); ctx.set_image_size(W,H)
hf = P.HuffmanFile(image); ctx.stream_begin(hf.header()); ctx.upload_batches(0,[hf.blob(b) for b in range(hf.numBatches)])
for rep in range(3):
    p = scenes.with_flags(scenes.cameras(W,H)["closeup"], lod_percent=100, cull=1)
    frame = pdist.SlicedFrame(ctx, W, H, dev, 1)
    frame.bind()
    pdist.render_basic_sharded(ctx, frame, p, 1, merge="sliced")
    torch.cuda.synchronize()
    ofb,_ = of.render_basic(p)
    merged = frame.gather_merged_framebuffer().cpu().numpy()
    m = np.where(merged == np.iinfo(np.int64).max, -1, merged).view(np.uint64)
    raw = frame.fb.cpu().numpy()[:ofb.size]; r2 = np.where(raw == np.iinfo(np.int64).max, -1, raw).view(np.uint64)
    rec = frame.recv.cpu().numpy()[:ofb.size]; r3 = np.where(rec == np.iinfo(np.int64).max, -1, rec).view(np.uint64)
    print("rep", rep, "merged diff", (m != ofb).sum(), "fb diff", (r2 != ofb).sum(), "recv diff", (r3 != ofb).sum(), "covered", (ofb != 2**64-1).sum())
    if (m != ofb).any():
        i = np.nonzero(m != ofb)[0]; print(len(i), i[:10], i[-3:], [hex(x) for x in m[i[:5]]], [hex(x) for x in ofb[i[:5]]])
    frame.release()
