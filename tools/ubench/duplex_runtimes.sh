# the duplex probe on the image's ROCm runtime (/opt/rocm) and on the one PyTorch bundles (torch/lib), then the torch microbenchmark on both
T=$(python -c "import torch,os; print(os.path.join(os.path.dirname(torch.__file__),'lib'))" 2>/dev/null)
mkdir -p /tmp/trt && ln -sf $T/libamdhip64.so /tmp/trt/libamdhip64.so.7 && ln -sf $T/libhsa-runtime64.so /tmp/trt/libhsa-runtime64.so.1
for f in $T/librocprofiler-register.so* $T/libamd_comgr.so* $T/libdrm*.so* $T/libnuma*.so* $T/libelf*.so*; do [ -e "$f" ] && ln -sf $f /tmp/trt/; done
echo "== /opt/rocm runtime"; timeout -k 10 100 tools/ubench/duplex 256 512
echo "== torch-bundled runtime"; LD_LIBRARY_PATH=/tmp/trt timeout -k 10 100 tools/ubench/duplex 256 512
echo "== tools/pcie_bw.py, torch's own runtime"; timeout -k 10 200 python tools/pcie_bw.py | grep -E "160MB|peak"
echo "== tools/pcie_bw.py, /opt/rocm runtime loaded first"; LD_PRELOAD=/opt/rocm/lib/libhsa-runtime64.so.1:/opt/rocm/lib/libamdhip64.so.7 timeout -k 10 200 python tools/pcie_bw.py | grep -E "160MB|peak"
