"""CPU oracle for the render-and-compare hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``foundationpose_amd/`` may import this
package; it is used by ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` as the checker / CPU baseline.

Parity status:
  * network definitions (``oracle.nets``): PINNED against outputs of the reference's own
    ``learning/models/*.py`` (tests/golden/nets_golden.npz, made by tests/golden/make_golden.py).
  * image-space ops (``oracle.ops`` / fp_oracle.c): "parity unpinned" -- they restate
    nvdiffrast / kornia / pytorch3d / warp semantics that are absent from /root/reference and
    from this container, and the reference ships no tests or golden vectors for them.
"""
