"""Per-box calibration for bench.py: what THIS MI355X sustains, measured in the same process as the benchmark.

The same binaries moved 7-8.6 % between two "identical" boxes (power-limited clocks): a rate quoted against the nominal 2.5 PFLOP/s cannot be
compared across rounds without a figure for the box. Two figures, both cheap:

  * mfma_sustained()  the fixed matrix-pipe microkernel of the library (yume_calibrate_mfma: one wave per SIMD on every CU issuing
                      v_mfma_f32_32x32x16_bf16 back to back on random operands, nothing else), run for a few hundred ms so that the power management settles:
                      the dense bf16 rate and the clock the chip holds under a max-toggle pure MFMA load (uniform-random sign / mantissa
                      bits; constant operands run 25-30 % faster: profiles/r5_calibrate_constant_operands.json). A yardstick for comparing
                      boxes, NOT an upper bound: real operands (Gaussian-like activations) toggle fewer bits than the calibration's;
  * gemm_reference()  one fixed launch of the product GEMM (8192^3, bf16 out): a reference launch with real operand traffic.
"""
import torch

from . import _lib, ops

FLOP_PER_MFMA = 2 * 32 * 32 * 16          # v_mfma_f32_32x32x16_bf16
CLK_PER_MFMA = 32                         # matrix-pipe clocks of one SIMD per instruction (MI355X_MICROARCH.md; 2.5 PF = 256 CUs x 4 x 2.4 GHz)


def mfma_sustained(device, settle_s=0.3, measure_s=0.4, iters=20000):
    """-> dict(tflops, clock_ghz, s_memtime_ghz, launches, ms_per_launch). clock_ghz = the matrix pipe's clock implied by the rate at 32
    clocks per instruction; s_memtime_ghz = shader-counter ticks per second of wall time as the kernel itself read them (s_memtime against
    the constant 100 MHz s_memrealtime) — a cross-check where that counter runs at the shader clock."""
    lib = _lib.load()
    dev = torch.device(device)
    ncu = torch.cuda.get_device_properties(dev).multi_processor_count
    with torch.cuda.device(dev):
        ticks = torch.zeros((ncu, 2), dtype=torch.int64, device=dev)
        sink = torch.zeros(1, dtype=torch.float32, device=dev)
        st = torch.cuda.current_stream().cuda_stream

        def launch():
            _lib.check(lib.yume_calibrate_mfma(iters, ncu, ticks.data_ptr(), sink.data_ptr(), st), "yume_calibrate_mfma")

        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        launch()                                  # untimed: first-use overhead (code object load) must not shrink the loops below
        torch.cuda.synchronize()
        e0.record()
        launch()
        e1.record()
        torch.cuda.synchronize()
        one = max(e0.elapsed_time(e1) * 1e-3, 1e-4)
        for _ in range(max(1, int(settle_s / one))):
            launch()
        n, dt = max(3, int(measure_s / one)), 0.0
        while True:                               # at least measure_s of MEASURED time (the clock drops as the load settles)
            e0.record()
            for _ in range(n):
                launch()
            e1.record()
            torch.cuda.synchronize()
            dt = e0.elapsed_time(e1) * 1e-3
            if dt >= 0.8 * measure_s or n >= 100000:
                break
            n = int(n * max(1.5, measure_s / max(dt, 1e-6))) + 1
        t = ticks.double().cpu()
    mfma_per_wave = iters * 16 * n
    rate = ncu * 4 * mfma_per_wave * FLOP_PER_MFMA / dt
    good = t[:, 1] > 0
    smt = float((t[good, 0] / (t[good, 1] / 100e6)).mean()) / 1e9 if bool(good.any()) else None
    return {"tflops": rate / 1e12, "clock_ghz": mfma_per_wave * CLK_PER_MFMA / dt / 1e9, "s_memtime_ghz": smt, "launches": n,
            "ms_per_launch": dt / n * 1e3, "cus": ncu, "measured_s": dt,
            "kernel": "yume_calibrate_mfma: 4 waves per CU x v_mfma_f32_32x32x16_bf16 back to back on random operands (32768 flop, 32 pipe clocks each), no memory traffic"}


def gemm_reference(device, n=8192, reps=20):
    """one fixed launch of the product GEMM (n^3, bf16 in / out, no bias): TFLOP/s over `reps` back-to-back launches."""
    dev = torch.device(device)
    g = torch.Generator(device=dev).manual_seed(99)
    a = (torch.randn((n, n), generator=g, device=dev) * 0.05).to(torch.bfloat16)
    w = (torch.randn((n, n), generator=g, device=dev) * 0.05).to(torch.bfloat16)
    out = torch.empty((n, n), dtype=torch.bfloat16, device=dev)
    with torch.cuda.device(dev):
        for _ in range(5):
            ops.gemm_bf16(a, w, None, out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.gemm_bf16(a, w, None, out)
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return {"tflops": 2.0 * n ** 3 / (ms * 1e-3) / 1e12, "ms_per_launch": ms, "shape": f"{n}x{n}x{n} bf16, yume_gemm_bf16 (gemm_w4_kernel)"}
