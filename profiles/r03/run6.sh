cd /root/repo
for hp in 0 f16; do DGCNN_HEAD_PLANES=$hp python profiles/r03/host_profile.py 2>&1 | grep -v amdgpu.ids | head -60; done
