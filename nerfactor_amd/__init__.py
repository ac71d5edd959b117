"""MI355X-native NeRFactor hot path: libnfx.so (include/nfx.h) behind the reference's plugin surface."""
import os

# One process per GPU over RCCL: on this driver stack device memory is shared between processes through dmabuf only (the legacy
# IPC mode makes RCCL's start-up fail with "hipIpcGetMemHandle: invalid argument").  The HIP runtime reads the variable when it
# starts, i.e. at the first device call — a default set here, at import, is in time; an exported value wins.
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
