"""Authoring tools: writers for the reference's .vks / .vkt / .lights / .rt_save formats and procedural scenes.

They take the place of the reference's Blender exporter (tools/io_export_vulkan_blender28.py), light exporter
(tools/io_export_spherical_lights_blender40.py) and texture converter (tools/texture_conversion/main.c) for
scenes that are generated instead of modelled.
"""
