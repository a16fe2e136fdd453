import torch, time
dev="cuda"
g=torch.Generator(device=dev).manual_seed(0)
for M in (1024,4096):
  for (K,N) in [(4096,6144),(4096,4096),(4096,28672),(14336,4096)]:
    a=((torch.rand(M,K,device=dev,generator=g)-0.5)*8).to(torch.float8_e4m3fn)
    w=((torch.rand(N,K,device=dev,generator=g)-0.5)*8).to(torch.float8_e4m3fn)
    sa=torch.rand(M,1,device=dev,generator=g)*1e-2+1e-3
    sb=torch.rand(1,N,device=dev,generator=g)*1e-2+1e-3
    try:
        f=lambda: torch._scaled_mm(a, w.t(), scale_a=sa, scale_b=sb, out_dtype=torch.bfloat16)
        for _ in range(3): f()
        torch.cuda.synchronize()
        st,en=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        st.record()
        for _ in range(20): f()
        en.record(); torch.cuda.synchronize()
        us=st.elapsed_time(en)*1e3/20
        print(f"rowwise M={M} K={K} N={N}: {us:.1f} us  {2*M*N*K/us/1e6:.0f} TFLOP/s")
    except Exception as e:
        print("rowwise failed", M,K,N, str(e)[:200])
        try:
            s1=torch.tensor(1.0,device=dev); 
            f=lambda: torch._scaled_mm(a, w.t(), scale_a=s1, scale_b=s1, out_dtype=torch.bfloat16)
            for _ in range(3): f()
            torch.cuda.synchronize()
            st,en=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
            st.record()
            for _ in range(20): f()
            en.record(); torch.cuda.synchronize()
            us=st.elapsed_time(en)*1e3/20
            print(f"tensorwise M={M} K={K} N={N}: {us:.1f} us  {2*M*N*K/us/1e6:.0f} TFLOP/s")
        except Exception as e2:
            print("tensorwise failed", str(e2)[:200])
