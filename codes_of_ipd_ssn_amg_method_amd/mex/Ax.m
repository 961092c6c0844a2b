function y = Ax(x,p,q)
% Drop-in shim with the reference's signature: forwards to libipdamg (HIP, gfx950)
% through the MEX gateway ipd_mex.  See INTEGRATION.md.
[y] = ipd_mex('Ax', x,p,q);
end
