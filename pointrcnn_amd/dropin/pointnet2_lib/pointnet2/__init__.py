"""`pointnet2_lib.pointnet2` -- same module names the reference imports
(lib/net/pointnet2_msg.py:3, lib/net/rcnn_net.py:4,6, lib/net/rpn.py:5)."""
