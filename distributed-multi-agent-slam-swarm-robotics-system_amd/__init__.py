"""MI355X-native central-mapper hot path (QuasarPacket ingest -> raycast -> loop closure ->
grid merge -> ZONE), a drop-in for that path of
deevinandu/Distributed-Multi-Agent-SLAM-Swarm-Robotics-System.

The directory name carries hyphens, so import it by string:
    importlib.import_module("distributed-multi-agent-slam-swarm-robotics-system_amd")
or through the alias module `quasar_amd` at the repository root.
"""
from . import protocol
from ._lib import QuasarError, QsConfig, build, load, LIB_PATH
from .mapper import OccupancyGrid, PoseGraphSLAM, QuasarMapper
from .protocol import (PACKET_FMT, PACKET_FMT_V1, PACKET_SIZE, PACKET_SIZE_V1, ZONE_FMT, ZONE_SIZE,
                       compute_bounding_box, pack_packet, pack_packets, zone_packet)

__all__ = ["protocol", "QuasarError", "QsConfig", "build", "load", "LIB_PATH", "OccupancyGrid",
           "PoseGraphSLAM", "QuasarMapper", "PACKET_FMT", "PACKET_FMT_V1", "PACKET_SIZE",
           "PACKET_SIZE_V1", "ZONE_FMT", "ZONE_SIZE", "compute_bounding_box", "pack_packet",
           "pack_packets", "zone_packet"]
