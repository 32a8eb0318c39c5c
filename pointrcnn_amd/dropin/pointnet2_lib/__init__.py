"""Drop-in for the reference's un-vendored `pointnet2_lib` git submodule (.gitmodules:1-4)."""
