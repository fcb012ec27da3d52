#!/bin/bash
# Host side of files -> FASTA on the GPU box's cores: buffer strategies of vc_hostbuf.h, first and later runs of a process.
cat /sys/kernel/mm/transparent_hugepage/enabled /sys/kernel/mm/transparent_hugepage/defrag; nproc
for m in 0 1 2; do
  echo "== VC_HOSTBUF=$m"
  VC_HOSTBUF=$m VC_FILES_HOST_ONLY=1 VC_IO_TIMING=${VC_IO_TIMING:-} python - <<'PY' 2>&1 | grep -v "^generated" | grep "files ->\|parse\|assembly\|load\|vc_io"
import sys; sys.path.insert(0,'tools')
import gpu_files_e2e as fe
for i in range(3):
    r=fe.main(200,10000,64,'/tmp/vc_files',python_too=False,quiet=False)
    r=None
PY
done
