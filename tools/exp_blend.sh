# Development: blend_probe.py under every library variant in exp/ (built by hand with -D switches).
for f in exp/libsxe_b*.so; do cp $f renderer_amd/libsxe_hip.so; echo $f; timeout -k 10 200 python tools/blend_probe.py 2>&1 | grep translucent || exit 1; done
