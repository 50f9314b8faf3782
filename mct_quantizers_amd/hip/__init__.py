"""gfx950 native layer: ctypes binding (native), launch/routing (ops), build helper (build)."""
