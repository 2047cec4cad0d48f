import sys, ctypes, torch
sys.path[:0]=['/root/repo/llm-inference_amd']
import cmhsa
lib=cmhsa.load_library()
dev=torch.device('cuda:0')
sp=ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for rows, cols in ((64*512,1600),(64*72,1600),(8*512,1600),(256,768),(512,1600),(4096,768)):
    x=torch.randn(rows,cols,device=dev); d=torch.randn(rows,cols,device=dev); w=torch.randn(cols,device=dev); b=torch.randn(cols,device=dev)
    s=torch.empty_like(x); y=torch.empty_like(x)
    a=(x.data_ptr(),d.data_ptr(),w.data_ptr(),b.data_ptr(),s.data_ptr(),y.data_ptr(),rows,cols,cols,cols,1e-5,sp)
    for _ in range(20): lib.gpt2_add_layernorm_f32_fwd(*a)
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): lib.gpt2_add_layernorm_f32_fwd(*a)
    e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)*1e3/50
    # torch pair
    ln=torch.nn.LayerNorm(cols).to(dev)
    for _ in range(20): ln(x+d)
    torch.cuda.synchronize(); e0.record()
    for _ in range(50): ln(x+d)
    e1.record(); torch.cuda.synchronize()
    print(rows, cols, f"fused {us:.2f} us = {rows*cols*16/us/1e3:.0f} GB/s; torch add+LN {e0.elapsed_time(e1)*1e3/50:.2f} us")
