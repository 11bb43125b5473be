import torch, time
dev = "cuda:0"
Nz, Ny, Nx = 512, 512, 512
rhs = torch.randn(Nz, Ny, Nx, dtype=torch.float64, device=dev)
def t(label, fn, reps=5):
    fn(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): out = fn()
    b.record(); torch.cuda.synchronize()
    print(f"{label:50s} {a.elapsed_time(b)/reps:8.3f} ms"); return out
R = t("rfft dim=2", lambda: torch.fft.rfft(rhs, dim=2))
S = t("fft dim=1 (strided)", lambda: torch.fft.fft(R, dim=1))
S2 = t("  .contiguous()", lambda: S.contiguous())
t("ifft dim=1", lambda: torch.fft.ifft(S2, dim=1))
t("irfft dim=2 n=Nx", lambda: torch.fft.irfft(S2[:, :, :257], n=Nx, dim=2))
Rt = t("transpose copy (Nz,Ny,K)->(Nz,K,Ny)", lambda: R.permute(0, 2, 1).contiguous())
t("fft dim=-1 contiguous on (Nz,K,Ny)", lambda: torch.fft.fft(Rt, dim=2))
t("fft2 dims (1,2) complex", lambda: torch.fft.fft2(R, dim=(1, 2)))
t("rfft2 dims (1,2)", lambda: torch.fft.rfft2(rhs, dim=(1, 2)))
f = torch.randn(518, 518, 518, dtype=torch.float64, device=dev)
def halo(fs):
    for x in fs:
        x[:, :3, :] = x[:, 512:515, :]; x[:, 515:, :] = x[:, 3:6, :]
t("12-field y self-halo copies", lambda: halo([f] * 12))
