#version 450
// Second pass of the capture: the id of the triangle that wins each pixel (primitives of a draw are numbered from 0
// in index order), with the reference's own vertex shader in front of it.
layout(location = 0) out uint outId;

void main() {
    outId = uint(gl_PrimitiveID);
}
