"""Host-side glue of the MI355X-native LSTM-baseline path: builds and binds libfsmg
(the C-ABI in include/fsmg.h).  There is no CPU fallback: without the HIP library or
without a gfx950 device every entry point raises."""
from fsmg.binding import FsmgError, FsmgModel, library_path, load_library   # noqa: F401
